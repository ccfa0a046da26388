// kernels_trait.hpp -- the TRAIT-GRANULAR loop of a drop-in caller (rsrl/examples/q_learning.rs:40-52), one C-ABI call per trait method:
//
//     t  = env.transition(a)        rsrl_hip_domain_step     rsrl_domains/src/lib.rs:436-446
//          agent.handle(&t)         rsrl_hip_handle          control/td/{q_learning.rs:51-71, sarsa.rs:53-75, expected_sarsa.rs:45-66}, pal.rs:35-60
//          terminal -> new episode  rsrl_hip_domain_reset    examples/q_learning.rs:37, :47-51
//     a' = policy.sample(rng, s')   rsrl_hip_policy_sample   policies/*.rs  (states = NULL: the ctx's own envs, env.emit().state())
//
// on the register-family Fourier bases with per-learner weights in the LEARNER-MAJOR layout (a ctx created with steps_per_launch = 1: W[N][A][F]),
// where Handler::handle is an HBM stream: 432 B of weights in, the 144 B of the touched column out.
//
//   k_trait_lm<.., TRAIT_HANDLE>  handle on caller-supplied transitions: the wave's 64 x A x F image by buffer_load ... lds, ONE pass over it for Q(s,.) and
//                                 Q(s',.), the touched column written back as whole sectors (the machinery of k_step_reg_lm, kernels_reg.hpp), and the
//                                 HAND-OVER: Q(s',.) under the UPDATED weights -- what the policy.sample that follows would have to stream W a second time
//                                 for -- left in the ctx's cache, keyed by the state it belongs to (s', or the new episode's s0 after a terminal transition).
//   k_trait_sample                policy.sample: a learner whose state is bit for bit the cache's key takes Q(s,.) from the cache (20 B instead of 432 B),
//                                 any other learner evaluates Q from its weights and refreshes its cache entry.
//   k_trait_lm<.., TRAIT_STEP>    the four calls of one batch-step in ONE launch (abi_trait.hip defers the calls of a ctx-owned stream and launches this
//                                 kernel when they arrive in the loop's order with device pointers; every output array of the separate calls is written).
//
// The hand-over is INVISIBLE in the results: it holds the bits a fresh evaluation yields (the untouched columns' dot products are unchanged by the update,
// the touched column's is recomputed from the updated column in the same summation order), so a cache hit and a miss return the same action values, and the
// loop is bit-identical to the oracle's reference-order loop (oracle/rsrl_oracle_impl.h orc_run_train, instantiation f32d) whichever kernels serve it
// (tests/test_gpu_trait_loop.py).  rsrl_hip_train's fused loops carry Q(s,.) with a rank-1 correction instead -- equal in exact arithmetic, not always in
// the last bit -- which is why the comparison is with the reference-order oracle and not with rsrl_hip_train.
#pragma once

#include "kernels_reg.hpp"

namespace rsrl {

enum { TRAIT_HANDLE = 1, TRAIT_STEP = 2 };

struct TraitIo {
    // TRAIT_HANDLE: the caller's Mn transitions (SoA, stride Mn); td_out optional
    const float* from; const int32_t* act; const float* rew; const float* to; const uint8_t* termf;
    float* td_out;
    // TRAIT_STEP: `act` = rsrl_hip_domain_step's actions argument (null: the ctx's pending actions); its outputs and policy_sample's (stride n_envs; each optional)
    float* o_from; float* o_to; float* o_rew; uint8_t* o_term; int32_t* o_act;
    float* qkey;          // [D][N]: the state qcache's entry of a learner belongs to
    int64_t Mn;
};

template <int D>
__device__ __forceinline__ bool same_bits(const float (&x)[D], const float (&y)[D]) {
    bool eq = true;
#pragma unroll
    for (int d = 0; d < D; ++d) eq &= (__builtin_bit_cast(uint32_t, x[d]) == __builtin_bit_cast(uint32_t, y[d])) & (x[d] == x[d]);   // (a NaN never hits: the host empties the cache with NaN keys)
    return eq;
}

// POLICY < 0: the behaviour policy's kind is read at run time (TRAIT_HANDLE: only SARSA / ExpectedSARSA look at it, through the agent's policy)
template <int DOMAIN, int ORDER, int ALGO, int POLICY, int MODE>
__global__ __launch_bounds__(kBlock) void k_trait_lm(Common c, TraitIo io, uint64_t t) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F, AF = A * F;
    static_assert(AF % 4 == 0 && F % 4 == 0, "16-byte rows");
    static_assert(A <= 3, "the column selects are laid out for A <= 3");
    constexpr int AF4 = AF / 4, F4 = F / 4;
    __shared__ __attribute__((aligned(16))) float lds[kBlock * AF];
    const int64_t N = c.n_envs;                                         // stride of the ctx's own arrays
    const int64_t Mn = MODE == TRAIT_HANDLE ? io.Mn : N;                // learners of this launch = stride of the caller's transition arrays
    const int lane = (int)(threadIdx.x & 63), wv_id = (int)(threadIdx.x >> 6);
    const int64_t wbase = (int64_t)blockIdx.x * kBlock + wv_id * 64;    // first learner of this wave
    const int64_t i = wbase + lane;
    float* __restrict__ img = lds + wv_id * 64 * AF;                    // this wave's 64 x A x F image
    const int64_t remain = Mn - wbase;
    const uint32_t img_bytes = (uint32_t)((remain < 64 ? (remain < 0 ? 0 : remain) : 64) * AF * 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(c.W + wbase * AF), 0, (int)img_bytes, 0x00020000);

    // the small loads the first arithmetic needs go out BEFORE the weight stream (vmcnt retires in issue order)
    const int64_t il = i < Mn ? i : Mn - 1;
    float s[D], ns[D];
    int a_raw; float r = 0.0f; bool term = false;
    if constexpr (MODE == TRAIT_HANDLE) {
#pragma unroll
        for (int d = 0; d < D; ++d) { s[d] = io.from[(int64_t)d * Mn + il]; ns[d] = io.to[(int64_t)d * Mn + il]; }
        a_raw = io.act[il]; r = io.rew[il]; term = io.termf[il] != 0;
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + il];
        a_raw = io.act ? io.act[il] : c.action[il];
    }
    const int a = clamp_action<A>(a_raw);                               // the weight column (a device array cannot be validated by the host: rsrl_hip.hip)
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef int i4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int m = 0; m < AF4; ++m)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(img + 64 * 4 * m), 16, lane * 16, 64 * 16 * m, 0, 0);
    // where this lane stores in the write-back of the touched columns: piece (g & 3) of sector t of learner j's column (k_step_reg_lm)
    constexpr int NSEC = (48 + F * 4 + 63) / 64;
    int wb_off[4 * NSEC];
    {
        const int sec = i < Mn ? (((lane * AF + a * F) * 4) & ~63) : 0x40000000;
#pragma unroll
        for (int p = 0; p < 4 * NSEC; ++p) {
            const int g = p * 64 + lane, sidx = g >> 2, j = sidx / NSEC, tt = sidx - j * NSEC;
            wb_off[p] = __builtin_amdgcn_ds_bpermute(j * 4, sec) + 64 * tt + 16 * (g & 3);
        }
    }

    // everything that needs only the transition runs underneath the weight stream
    PolicyParams pol = c.pol;
    if constexpr (POLICY >= 0) pol.kind = POLICY;
    AlgoParams alg = c.alg; alg.kind = ALGO;
    const uint32_t gid = (uint32_t)(c.env_offset + il);
    [[maybe_unused]] float ns_obs[D];                                    // TRAIT_STEP: the observed s' (terminal or not), rsrl_hip_domain_step's `next_states`
    if constexpr (MODE == TRAIT_STEP) {
        // ---- Domain::transition
#pragma unroll
        for (int d = 0; d < D; ++d) ns[d] = s[d];
        term = Dom::step(ns, a_raw, r);
#pragma unroll
        for (int d = 0; d < D; ++d) ns_obs[d] = ns[d];
    }
    // the state the hand-over is for: s', or -- after a terminal transition -- the new episode's s0 (what policy.sample sees next)
    if (term) Dom::reset(ns);
    float phi_s[F], phi_n[F], q_n[A];
    Bas::project(s, phi_s);
    Bas::project(ns, phi_n);
    U4 xin = U4{0, 0, 0, 0};
    if constexpr (ALGO == ALG_SARSA) xin = draw(c.seed, gid, t, BLK_INNER);
    [[maybe_unused]] U4 x = U4{0, 0, 0, 0};
    if constexpr (MODE == TRAIT_STEP) x = draw(c.seed, gid, t, BLK_STEP);

    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0): the issuing wave's covering wait orders its ds_reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // the image is private to this wave: no block barrier
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float wv[A][F];
#pragma unroll
    for (int k = 0; k < AF4; ++k) {
        const f4 v = *reinterpret_cast<const f4*>(img + lane * AF + 4 * k);
        wv[(4 * k) / F][(4 * k) % F] = v.x; wv[(4 * k + 1) / F][(4 * k + 1) % F] = v.y;
        wv[(4 * k + 2) / F][(4 * k + 2) % F] = v.z; wv[(4 * k + 3) / F][(4 * k + 3) % F] = v.w;
    }

    // ---- Handler::handle: Q(s,.) and Q(s',.) from the one image (pre-update weights)
    float qs_arr[A];
    q_from_reg<A, F>(wv, phi_s, qs_arr);
    const float qsa = (a == 0) ? qs_arr[0] : ((a == 1) ? qs_arr[A > 1 ? 1 : 0] : qs_arr[A > 2 ? 2 : 0]);
    q_from_reg<A, F>(wv, phi_n, q_n);
    float e, delta;
    if constexpr (ALGO == ALG_PAL) delta = td_error_pal<A>(alg, qs_arr, q_n, a, r, term, e);
    else delta = td_error_ap<A, ALGO>(alg, pol, c, qsa, q_n, r, term, xin, e);
    // ---- W[:,a] += lr * e * phi(s): old column from the LDS image (lane-dependent address), merged back into it
    const float scale = alg.lr * e;
    float one[1][F];
    float* __restrict__ colp = img + lane * AF + a * F;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const f4 o = *reinterpret_cast<const f4*>(colp + 4 * k);
        f4 v;
        v.x = fmaf(scale, phi_s[4 * k], o.x); v.y = fmaf(scale, phi_s[4 * k + 1], o.y);
        v.z = fmaf(scale, phi_s[4 * k + 2], o.z); v.w = fmaf(scale, phi_s[4 * k + 3], o.w);
        one[0][4 * k] = v.x; one[0][4 * k + 1] = v.y; one[0][4 * k + 2] = v.z; one[0][4 * k + 3] = v.w;
        *reinterpret_cast<f4*>(colp + 4 * k) = v;
    }
    // the touched columns go back as WHOLE 64-byte sectors, four consecutive lanes per sector (k_step_reg_lm: 9.05 -> 7.94 us per launch)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    static_assert((64 * AF * 4) % 64 == 0, "no sector is shared between two waves' images");
#pragma unroll
    for (int p = 0; p < 4 * NSEC; ++p) {
        const f4 v = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(img) + (wb_off[p] & 0xffff));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, wb_off[p], 0, 0);
    }
    // ---- the hand-over: Q(s',.) under the UPDATED weights = what a fresh evaluation returns, bit for bit (only column a changed, and its dot product
    //      is taken from the updated column in q_from_reg's summation order)
    {
        float qa[1];
        q_from_reg<1, F>(one, phi_n, qa);
#pragma unroll
        for (int b = 0; b < A; ++b) q_n[b] = (a == b) ? qa[0] : q_n[b];
    }
    if (i < Mn) {
        if constexpr (MODE == TRAIT_HANDLE) {
            if (io.td_out) io.td_out[i] = delta;
        } else {
            // ---- the transition itself: rsrl_hip_domain_step's outputs (stored here, behind the weight stream: stores count in vmcnt)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (io.o_from) io.o_from[(int64_t)d * N + i] = s[d];
                if (io.o_to) io.o_to[(int64_t)d * N + i] = ns_obs[d];
            }
            if (io.o_rew) io.o_rew[i] = r;
            if (io.o_term) io.o_term[i] = term ? 1 : 0;
            // ---- the new episode of a finished one (rsrl_hip_domain_reset with the terminal flags as its mask) and policy.sample at the ctx's own state
            const int na = policy_sample<A>(pol, q_n, x);
#pragma unroll
            for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = ns[d];
            c.action[i] = na;
            if (io.o_act) io.o_act[i] = na;
            if (term) c.ep_step[i] = 0;
            if (io.td_out) io.td_out[i] = delta;
        }
#pragma unroll
        for (int b = 0; b < A; ++b) c.qcache[(int64_t)b * N + i] = q_n[b];
#pragma unroll
        for (int d = 0; d < D; ++d) io.qkey[(int64_t)d * N + i] = ns[d];
    }
}

// Policy::sample on the register-family Fourier bases, learner-major per-learner weights.  states == nullptr: the ctx's own envs (Domain::emit), the action
// also becomes the ctx's pending one; blk = BLK_STEP (the driver loop's behaviour draw of batch-step t) or BLK_API (the API stream, t = call counter).
template <int DOMAIN, int ORDER>
__global__ __launch_bounds__(kBlock) void k_trait_sample(Common c, const float* __restrict__ states, int64_t Mn, uint64_t t, uint32_t blk,
                                                          float* __restrict__ qkey, int32_t* __restrict__ actions_out) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F, AF = A * F;
    static_assert(AF % 4 == 0, "16-byte rows");
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    float s[D], key[D], q[A];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        s[d] = states ? states[(int64_t)d * Mn + i] : c.state[(int64_t)d * N + i];
        key[d] = qkey[(int64_t)d * N + i];
    }
#pragma unroll
    for (int b = 0; b < A; ++b) q[b] = c.qcache[(int64_t)b * N + i];
    if (!same_bits<D>(s, key)) {
        // no hand-over for this state (a restart the caller made by other means, a state of the caller's own): evaluate from the learner's weights
        typedef float f4 __attribute__((ext_vector_type(4)));
        float phi[F], wv[A][F];
        Bas::project(s, phi);
        const f4* __restrict__ wp = reinterpret_cast<const f4*>(c.W + i * AF);
#pragma unroll
        for (int k = 0; k < AF / 4; ++k) {
            const f4 v = wp[k];
            wv[(4 * k) / F][(4 * k) % F] = v.x; wv[(4 * k + 1) / F][(4 * k + 1) % F] = v.y;
            wv[(4 * k + 2) / F][(4 * k + 2) % F] = v.z; wv[(4 * k + 3) / F][(4 * k + 3) % F] = v.w;
        }
        q_from_reg<A, F>(wv, phi, q);
        // the entry now belongs to this state
#pragma unroll
        for (int b = 0; b < A; ++b) c.qcache[(int64_t)b * N + i] = q[b];
#pragma unroll
        for (int d = 0; d < D; ++d) qkey[(int64_t)d * N + i] = s[d];
    }
    const U4 x = draw(c.seed, (uint32_t)(c.env_offset + i), t, blk);
    const int a = policy_sample<A>(c.pol, q, x);
    actions_out[i] = a;
    if (!states) c.action[i] = a;
}

}  // namespace rsrl

// launchers (one translation unit per group of instantiations: trait_d0a.hip, trait_d0b.hip, trait_d12.hip); false = no instantiation
namespace rsrl {
// policy < 0: Handler::handle on the caller's transitions (TRAIT_HANDLE); otherwise the fused batch-step (TRAIT_STEP)
bool launch_trait_lm(int domain, int order, int algo, int policy, hipStream_t st, const Common& k, const TraitIo& io, uint64_t t);
bool launch_trait_sample(int domain, int order, hipStream_t st, const Common& k, const float* states, int64_t Mn, uint64_t t, uint32_t blk, float* qkey,
                         int32_t* actions_out);
bool trait_lm_available(int domain, int order, int algo);
}  // namespace rsrl

#define RSRL_TRAIT_CASE(DM, OR, AL)                                                                                                              \
    if (domain == DM && order == OR && algo == AL) {                                                                                             \
        const dim3 grid((unsigned)(((policy < 0 ? io.Mn : k.n_envs) + kBlock - 1) / kBlock)), block(kBlock);                                   \
        switch (policy) {                                                                                                                        \
        case 0: hipLaunchKernelGGL((k_trait_lm<DM, OR, AL, 0, TRAIT_STEP>), grid, block, 0, st, k, io, t); break;                              \
        case 1: hipLaunchKernelGGL((k_trait_lm<DM, OR, AL, 1, TRAIT_STEP>), grid, block, 0, st, k, io, t); break;                              \
        case 2: hipLaunchKernelGGL((k_trait_lm<DM, OR, AL, 2, TRAIT_STEP>), grid, block, 0, st, k, io, t); break;                              \
        case 3: hipLaunchKernelGGL((k_trait_lm<DM, OR, AL, 3, TRAIT_STEP>), grid, block, 0, st, k, io, t); break;                              \
        default: hipLaunchKernelGGL((k_trait_lm<DM, OR, AL, -1, TRAIT_HANDLE>), grid, block, 0, st, k, io, t); break;                          \
        }                                                                                                                                        \
        return true;                                                                                                                             \
    }
#define RSRL_TRAIT_ALGOS(DM, OR) RSRL_TRAIT_CASE(DM, OR, 0) RSRL_TRAIT_CASE(DM, OR, 1) RSRL_TRAIT_CASE(DM, OR, 2) RSRL_TRAIT_CASE(DM, OR, 5)
#define RSRL_TRAIT_SAMPLE_CASE(DM, OR)                                                                                                           \
    if (domain == DM && order == OR) {                                                                                                           \
        hipLaunchKernelGGL((k_trait_sample<DM, OR>), dim3((unsigned)((Mn + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, k, states, Mn, t, blk, qkey, actions_out); \
        return true;                                                                                                                             \
    }
