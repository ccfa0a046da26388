// kernels_reg_q4.hpp -- the single-step streaming kernel with FOUR LANES PER LEARNER (round 3).
//
// k_step_reg_lm (kernels_reg.hpp) gives every learner one lane: a wave owns 64 learners and a 27 KB LDS image (A*F = 108 weights
// each), a block of four waves fills the CU's LDS budget, and every SIMD holds ONE wave whose three phases -- load the image,
// compute, store the touched column -- cannot overlap with anything.  Here a learner is a QUAD of lanes: lane b of the quad owns
// action column b (F weights), so a wave covers 16 learners, its image is 6.9 KB, and up to sixteen waves per CU are resident.
// Measured (us per launch, one-lane kernel -> this one): 65 536 learners 9.05 -> 9.5 (ONE round of waves either way, all in the
// same phase at the same time, and the quad replicates the learner's scalar work in four lanes: the one-lane kernel stays the
// default there), 131 072: 21.3 -> 19.3, 262 144: 38.0 -> 31.5 (0.52 -> 0.62 of 8 TB/s on the 608 B/env-step accounting),
// 1 M: 160-175 -> 152 -- where the access pattern alone, with no arithmetic, takes 127-140 (scripts/ubench/stream_pattern.hip).
//   load : the wave's 16 x A x F image as ceil(432 / 64) fully coalesced 16-B loads per lane (contiguous 6.9 KB), through LDS
//          (linear 16-B writes, then lane (learner q, column b) reads its F weights: F/4 ds_read_b128)
//   math : transition, phi(s), phi(s'), the draws -- per quad, replicated in its four lanes (same instructions, same bits);
//          Q(s', b) = <W[:, b], phi(s')> in lane b, the A values exchanged inside the quad by DPP quad_perm broadcasts;
//          the TD error and the policy in every lane
//   store: lane a of the quad (a = the action taken) updates and stores its column: F/4 16-B stores, F*4 contiguous bytes
// Same helpers, same operation order per element as k_step_reg_lm / k_train_reg: bit-identical results (the tests compare the
// three).  Layout precondition as k_step_reg_lm: learner-major rows W[N][A][F], F % 4 == 0; and A <= 4.
#pragma once
#include "kernels_reg.hpp"

namespace rsrl {

// (cache-policy bits of the image loads / column stores stay 0: nt is SLOWER at every size -- loads nt 13.3 vs 9.8 us at 65 536 learners, W lives in
// L2 / MALL between launches; both nt 331 vs 168 at 1 M)

template <int J>
__device__ __forceinline__ float quad_bcast(float x) {                  // lane J of every quad, to the quad's four lanes
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), J * 0x55, 0xf, 0xf, false));
}

template <int DOMAIN, int ORDER, int ALGO, int POLICY>
__global__ __launch_bounds__(kBlock) void k_step_reg_q4(Common c, uint64_t t, DevStats* __restrict__ stats, const uint64_t* __restrict__ t_dev) {
    if (t_dev) t += *t_dev;
    if (c.dyn) { c.pol = c.dyn->pol; c.apol = c.dyn->apol; }
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F, AF = A * F;
    static_assert(F % 4 == 0 && A <= 4 && A <= 3, "16-byte columns, one lane of the quad per action, carried Q for A <= 3");
    constexpr int LPW = 16, IMG = LPW * AF, IMG4 = IMG / 4, NLD = (IMG4 + 63) / 64, F4 = F / 4;
    constexpr int IMGP = NLD * 64 * 4;                                   // image stride in LDS: whole load instructions (the last one overhangs)
    __shared__ __attribute__((aligned(16))) float lds[(kBlock / 64) * IMGP];
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef int i4 __attribute__((ext_vector_type(4)));
    const int64_t N = c.n_envs;
    const int lane = (int)(threadIdx.x & 63), wv_id = (int)(threadIdx.x >> 6);
    const int q = lane >> 2, b = lane & 3;
    const int64_t wbase = ((int64_t)blockIdx.x * (kBlock / 64) + wv_id) * LPW;       // first learner of this wave
    const int64_t i = wbase + q;
    float* __restrict__ img = lds + wv_id * IMGP;
    const int64_t remain = N - wbase;
    const uint32_t img_bytes = (uint32_t)((remain < LPW ? (remain < 0 ? 0 : remain) : LPW) * AF * 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(c.W + wbase * AF), 0, (int)img_bytes, 0x00020000);

    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;

    const int64_t il = i < N ? i : N - 1;
    float s[D], ns[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + il];
    const int a = c.action[il];
    uint32_t ep = c.ep_step[il] + 1;
    float qc0 = 0.0f, qc1 = 0.0f, qc2 = 0.0f;
    if (c.q_valid) {
        qc0 = c.qcache[il];
        qc1 = c.qcache[N + il];
        if constexpr (A > 2) qc2 = c.qcache[2 * N + il];
    }
#pragma unroll
    for (int m = 0; m < NLD; ++m)         // straight into the wave's LDS image (wave-uniform LDS base + lane * 16); beyond the image: zeros into the pad
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(img + 64 * 4 * m), 16, lane * 16, 64 * 16 * m, 0, 0);

    PolicyParams pol = c.pol; pol.kind = POLICY;
    AlgoParams alg = c.alg; alg.kind = ALGO;
    const uint32_t gid = (uint32_t)(c.env_offset + il);
    const uint32_t cap = c.max_episode_steps;
#pragma unroll
    for (int d = 0; d < D; ++d) ns[d] = s[d];
    float r;
    const bool term = Dom::step(ns, a, r);
    const bool trunc = !term && cap > 0 && ep >= cap;
    if (term) Dom::reset(ns);
    float phi_s[F], phi_n[F], q_n[A];
    Bas::project(s, phi_s);
    Bas::project(ns, phi_n);
    U4 xin = U4{0, 0, 0, 0};
    if constexpr (ALGO == ALG_SARSA) xin = draw(c.seed, gid, t, BLK_INNER);
    const U4 x = draw(c.seed, gid, t, BLK_STEP);

    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0): the issuing wave's covering wait orders its ds_reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // the image is private to this wave: no block barrier
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int bc = b < A ? b : 0;                                        // lane 3 of a three-action quad reads column 0 and is ignored
    [[maybe_unused]] const int col_off = (q * AF + bc * F) * 4;          // bytes, inside the wave's image and inside its slice of W
    float wcol[1][F];
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const f4 v = *reinterpret_cast<const f4*>(img + q * AF + bc * F + 4 * k);
        wcol[0][4 * k] = v.x; wcol[0][4 * k + 1] = v.y; wcol[0][4 * k + 2] = v.z; wcol[0][4 * k + 3] = v.w;
    }
    // A column dot products, one per lane of the quad, then shared by the quad
    auto quad_q = [&](const float (&col)[1][F], const float (&phi)[F], float (&out)[A]) {
        float qb[1];
        q_from_reg<1, F>(col, phi, qb);
        out[0] = quad_bcast<0>(qb[0]);
        if constexpr (A > 1) out[A > 1 ? 1 : 0] = quad_bcast<1>(qb[0]);
        if constexpr (A > 2) out[A > 2 ? 2 : 0] = quad_bcast<2>(qb[0]);
    };
    constexpr int P = RSRL_DOT_SPLIT;
    float qs_arr[A];
    if (c.q_valid) {
        qs_arr[0] = qc0;
        if constexpr (A > 1) qs_arr[1] = qc1;
        if constexpr (A > 2) qs_arr[2] = qc2;
    } else {
        quad_q(wcol, phi_s, qs_arr);
    }
    const float qsa = (a == 0) ? qs_arr[0] : ((a == 1) ? qs_arr[A > 1 ? 1 : 0] : qs_arr[A > 2 ? 2 : 0]);
    quad_q(wcol, phi_n, q_n);                                          // Q(s',.) with the PRE-update weights
    float e, delta;
    if constexpr (ALGO == ALG_PAL) delta = td_error_pal<A>(alg, qs_arr, q_n, a, r, term, e);
    else delta = td_error_ap<A, ALGO>(alg, pol, c, qsa, q_n, r, term, xin, e);
    // ---- W[:,a] += lr * e * phi(s): lane a of the quad
    const float scale = alg.lr * e;
    const bool mine = b == a;
    if (mine) {
#pragma unroll
        for (int k = 0; k < F4; ++k) {
            f4 v;
            v.x = fmaf(scale, phi_s[4 * k], wcol[0][4 * k]); v.y = fmaf(scale, phi_s[4 * k + 1], wcol[0][4 * k + 1]);
            v.z = fmaf(scale, phi_s[4 * k + 2], wcol[0][4 * k + 2]); v.w = fmaf(scale, phi_s[4 * k + 3], wcol[0][4 * k + 3]);
            wcol[0][4 * k] = v.x; wcol[0][4 * k + 1] = v.y; wcol[0][4 * k + 2] = v.z; wcol[0][4 * k + 3] = v.w;
            *reinterpret_cast<f4*>(img + q * AF + bc * F + 4 * k) = v;                   // merged into the wave's image, written back below
        }
    }
    // The touched column goes back as WHOLE 64-byte sectors, one store instruction per sector with the quad's four lanes writing
    // its four 16-byte pieces: F*4 = 144 bytes at a 16-byte-aligned offset dirty three sectors, and a partially written sector is a
    // read-modify-write for the memory side (scripts/ubench/stream_pattern.hip: read 432 + write 144 per learner, no arithmetic,
    // 6.2 us per launch at 65 536 learners and 140 us at 1 M; 4.7 and 127-135 us with the columns widened to whole sectors /
    // lines).  The bytes around the column come from the wave's image in LDS, where the quads' updates have been merged: two quads
    // whose sectors overlap store the same bytes.  The image is a multiple of 64 bytes long: no sector is shared between waves.
    // Measured on the kernel: 9.9 -> 9.5 us at 65 536 learners, 32.7 -> 31.5 at 262 144, 159 -> 152 at 1 M.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Round 5: once the weights no longer fit the 256 MiB Infinity Cache the column goes back as whole 128-BYTE LINES instead (two lines = four
    // store instructions per quad instead of three sectors): the bare pattern takes 126 us against 135 at 1 M learners but 25.4 against 23.7 at
    // 262 144, where the stores end in the cache (profiles/r03_ubench_stream_pattern.txt) -- so the width follows the working set.
    if (i < N) {
        static_assert((LPW * AF * 4) % 128 == 0, "no sector or line is shared between two waves' images");
        const bool lines = N * (int64_t)(AF * 4) > ((int64_t)256 << 20);              // (wave-uniform: a kernel argument)
        constexpr int NSEC = (48 + F * 4 + 63) / 64, NLINE2 = 2 * ((112 + F * 4 + 127) / 128);
        const int sec = ((q * AF + a * F) * 4) & (lines ? ~127 : ~63);
        const int n_st = lines ? NLINE2 : NSEC;
#pragma unroll
        for (int p = 0; p < (NLINE2 > NSEC ? NLINE2 : NSEC); ++p) {
            if (p < n_st) {
                const int off = sec + 64 * p + 16 * b;
                const f4 v = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(img) + off);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, off, 0, 0);
            }
        }
    }
    // ---- Q(s',.) with the UPDATED weights: only column a changed (rank-1 term, as k_step_reg_lm)
    {
        float dacc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) dacc[p] = 0.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) dacc[f % P] = fmaf(phi_s[f], phi_n[f], dacc[f % P]);
        const float dot = combine_partials<P>(dacc);
#pragma unroll
        for (int bb = 0; bb < A; ++bb) q_n[bb] = (a == bb) ? fmaf(scale, dot, q_n[bb]) : q_n[bb];
    }
    int na = policy_sample<A>(pol, q_n, x);
    sum_abs = (double)fabsf(delta); sum_r = (double)r;
    if (term) { n_ep = 1; sum_len = ep; ep = 0; }
    if (trunc) {                               // step cap: new episode needs Q(s0) with the updated W (whole quads take this branch)
        n_ep = 1; n_trunc = 1; sum_len = ep; ep = 0;
        Dom::reset(ns);
        Bas::project(ns, phi_n);
        quad_q(wcol, phi_n, q_n);              // wcol is the updated column in lane a, the untouched one elsewhere
        const U4 xr = draw(c.seed, gid, t, BLK_RESET);
        na = policy_sample<A>(pol, q_n, xr);
    }
    if (i < N && b == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = ns[d];
        c.action[i] = na;
        c.ep_step[i] = ep;
#pragma unroll
        for (int bb = 0; bb < A; ++bb) c.qcache[(int64_t)bb * N + i] = q_n[bb];
    } else {
        n_ep = 0; n_trunc = 0; sum_len = 0; sum_abs = 0.0; sum_r = 0.0;          // one lane per learner reports
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

}  // namespace rsrl
