// kernels_reg.hpp -- "register family" kernels: one thread per learner, Fourier features
// (F <= 64) and, in the fused driver loop, the learner's whole weight matrix held in
// VGPRs for the duration of the launch.
//
// HBM layout (SoA, learner index fastest => every wave access is one contiguous line):
//   state  f32[D][N]      action i32[N]      ep_step u32[N]
//   W      f32[A][F][N]   (per-env mode)     W f32[A][F] (shared mode, Nw = 1)
#pragma once

#include "device_core.hpp"

namespace rsrl {

struct DevStats {
    unsigned long long env_steps, episodes, episodes_truncated, sum_episode_steps;
    double sum_abs_td_error, sum_reward;
};

// the parameters a driver changes between calls (EpsilonGreedy.epsilon is a pub field the reference's drivers decay,
// examples/sarsa_lambda.rs:68): graph nodes read them from device memory so that a change does not re-capture the graph
struct DynParams { PolicyParams pol, apol; };
struct Common {
    const DynParams* dyn;  // non-null inside a captured graph: pol / apol below are taken from here at run time
    int64_t n_envs;        // learners in this ctx
    int64_t env_offset;    // global id of learner 0
    uint64_t seed;
    PolicyParams pol;      // behaviour policy (the driver's policy.sample / mode)
    PolicyParams apol;     // the policy owned by the agent (SARSA's inner draw, ExpectedSARSA's expectation): sarsa.rs:35-41, expected_sarsa.rs:22-29
    int apol_same;         // apol is the behaviour policy object itself (what the reference examples build with make_shared)
    AlgoParams alg;
    uint32_t max_episode_steps;
    float* state;          // [D][N]
    int32_t* action;       // [N]
    uint32_t* ep_step;     // [N]
    float* W;              // [A][F][Nw]
    int64_t w_stride;      // stride between consecutive (action, feature) rows: n_envs (feature-major), 1 (shared, learner-major)
    int64_t w_ls;          // stride between learners: 1 (feature-major / shared), A*F (learner-major, the single-step layout)
    int shared;            // one approximator for all learners (weight_mode == RSRL_W_SHARED)
    float* qcache;         // [A][N] Q(s,.) of the CURRENT state with the current weights, carried between launches
    int q_valid;           // 0: qcache is stale (weights/states were changed from outside) -> recompute from W
    float* eps;            // [N] or null.  Non-null: EpsilonGreedy.epsilon is a field of every LEARNER (as it is of the reference's one
                           // learner, epsilon_greedy.rs:19) and the driver-loop kernels run the reference drivers' schedule on it --
                           // eps_i <- max(eps_i * eps_decay, eps_min) at every episode end of learner i (examples/sarsa_lambda.rs:68)
    float eps_decay, eps_min;
    int64_t xdelta;        // multi-rank peer exchange: (exchanges this ctx has performed on its receive buffer) - (batch-step counter).
                           // Slot parity and granule tags follow t + xdelta, a sequence number that only ever grows, whatever happens
                           // to the batch-step counter (a restored checkpoint sets it back; other exchange paths advance it alone)
};

constexpr int kBlock = 256;

// ---- EpsilonGreedy.epsilon per learner (Common::eps).  The reference's driver decays the pub field once per episode of its one
// learner, AFTER the episode's last handle / sample and BEFORE the next episode's initial sample (examples/sarsa_lambda.rs:48-75,
// :68); vectorised, every learner carries its own value.  fp32 on the device (eps_i * decay rounded once per episode; the f64
// oracle keeps the reference's f64 product); gen_bool's threshold is taken from it exactly as make_common takes it from the config.
__device__ __forceinline__ uint32_t eps_threshold(float eps) {
    const float v = eps * 16777216.0f;                                 // exact (a power of two)
    return v <= 0.0f ? 0u : (v >= 16777216.0f ? 16777216u : (uint32_t)v);
}
__device__ __forceinline__ void learner_eps_load(const Common& c, int64_t i, PolicyParams& pol) {
    if (c.eps) { const float e = c.eps[i]; pol.eps = e; pol.eps_thr = eps_threshold(e); }
}
// at the end of a step: the episode ended (terminal or step cap) -> the learner's epsilon decays; `pol` then samples the new episode's
// first action with the new value
__device__ __forceinline__ void learner_eps_step(const Common& c, bool episode_ended, PolicyParams& pol) {
    const float e2 = fmaxf(pol.eps * c.eps_decay, c.eps_min);
    pol.eps = episode_ended ? e2 : pol.eps;
    pol.eps_thr = eps_threshold(pol.eps);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-level reduction of the per-thread statistics into stats[blockIdx.x]
__device__ __forceinline__ void block_stats_accumulate(DevStats* __restrict__ stats, unsigned long long n_ep,
                                                       unsigned long long n_trunc, unsigned long long sum_len,
                                                       double sum_abs, double sum_r) {
    __shared__ unsigned long long sh_u[3][16];          // up to 1024 threads per block
    __shared__ double sh_d[2][16];
    n_ep = wave_sum(n_ep); n_trunc = wave_sum(n_trunc); sum_len = wave_sum(sum_len);
    sum_abs = wave_sum(sum_abs); sum_r = wave_sum(sum_r);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh_u[0][wave] = n_ep; sh_u[1][wave] = n_trunc; sh_u[2][wave] = sum_len;
        sh_d[0][wave] = sum_abs; sh_d[1][wave] = sum_r;
    }
    __syncthreads();
    if (threadIdx.x == 0 && stats) {
        DevStats acc = stats[blockIdx.x];
        for (int w = 0; w < nw; ++w) {
            acc.episodes += sh_u[0][w]; acc.episodes_truncated += sh_u[1][w]; acc.sum_episode_steps += sh_u[2][w];
            acc.sum_abs_td_error += sh_d[0][w]; acc.sum_reward += sh_d[1][w];
        }
        stats[blockIdx.x] = acc;
    }
}

// Dot products run as RSRL_DOT_SPLIT interleaved partial sums (acc_p takes the features f = p mod P in
// ascending order; q = (acc0 + acc1) + (acc2 + acc3)): a lone wave per SIMD cannot hide the latency of
// one 36-long dependent fma chain per action, 4 x 3 independent chains can.
#ifndef RSRL_DOT_SPLIT
#define RSRL_DOT_SPLIT 4
#endif
template <int P>
__device__ __forceinline__ float combine_partials(const float (&acc)[P]) {
    if constexpr (P == 1) return acc[0];
    else if constexpr (P == 2) return acc[0] + acc[1];
    else { static_assert(P == 4, "RSRL_DOT_SPLIT must be 1, 2 or 4"); return (acc[0] + acc[1]) + (acc[2] + acc[3]); }
}

// Q(s,.) = W^T phi(s) with phi in registers and W streamed from memory (learner-fastest layout)
//   Function<(S,)>::evaluate for VectorLFA        fa/linear.rs:303-311
template <int A, int F>
__device__ __forceinline__ void q_from_mem(const float* __restrict__ W, int64_t stride, int64_t wi,
                                           const float (&phi)[F], float (&q)[A]) {
    constexpr int P = RSRL_DOT_SPLIT;
#pragma unroll
    for (int b = 0; b < A; ++b) {
        float acc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = 0.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f % P] = fmaf(phi[f], W[((int64_t)(b * F + f)) * stride + wi], acc[f % P]);
        q[b] = combine_partials<P>(acc);
    }
}
template <int A, int F>
__device__ __forceinline__ void q_from_reg(const float (&w)[A][F], const float (&phi)[F], float (&q)[A]) {
    constexpr int P = RSRL_DOT_SPLIT;
#pragma unroll
    for (int b = 0; b < A; ++b) {
        float acc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = 0.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f % P] = fmaf(phi[f], w[b][f], acc[f % P]);
        q[b] = combine_partials<P>(acc);
    }
}
// ---- packed-pair variants (v_pk_fma_f32): at one wave per SIMD a wave issues one VALU instruction per ~3.4 cycles whatever
// its width, and a packed FMA (two IEEE fmas) costs ~5.2 -- 25 % fewer cycles for the dot products, the column update and
// the rank-1 term.  Same chain assignment as the scalar code (acc[f % 4] takes feature f), so results are bit-identical.
#ifndef RSRL_PK
#define RSRL_PK 1
#endif
template <int A, int H>
__device__ __forceinline__ void q_from_reg2(const f2 (&w)[A][H], const f2 (&phi)[H], float (&q)[A]) {
    static_assert(RSRL_DOT_SPLIT == 4 && H % 2 == 0, "pairs (f, f+1) with f = 0 mod 4 -> chains 0,1; f = 2 mod 4 -> chains 2,3");
#pragma unroll
    for (int b = 0; b < A; ++b) {
        f2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < H; j += 2) {
            a01 = __builtin_elementwise_fma(phi[j], w[b][j], a01);
            a23 = __builtin_elementwise_fma(phi[j + 1], w[b][j + 1], a23);
        }
        q[b] = (a01.x + a01.y) + (a23.x + a23.y);
    }
}
template <int H>
__device__ __forceinline__ float dot2(const f2 (&x)[H], const f2 (&y)[H]) {
    f2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < H; j += 2) {
        a01 = __builtin_elementwise_fma(x[j], y[j], a01);
        a23 = __builtin_elementwise_fma(x[j + 1], y[j + 1], a23);
    }
    return (a01.x + a01.y) + (a23.x + a23.y);
}
template <int F>
__device__ __forceinline__ void pack2(const float (&x)[F], f2 (&y)[F / 2]) {
#pragma unroll
    for (int j = 0; j < F / 2; ++j) y[j] = f2{x[2 * j], x[2 * j + 1]};
}

// Register-resident phi / W of one learner, scalar or packed-pair storage behind one interface.
template <int F, bool PK> struct PhiBuf {
    float v[F];
    __device__ __forceinline__ void set(const float (&x)[F]) {
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = x[f];
    }
    __device__ __forceinline__ static float dot(const PhiBuf& x, const PhiBuf& y) {
        constexpr int P = RSRL_DOT_SPLIT;
        float dacc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) dacc[p] = 0.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) dacc[f % P] = fmaf(x.v[f], y.v[f], dacc[f % P]);
        return combine_partials<P>(dacc);
    }
};
template <int F> struct PhiBuf<F, true> {
    f2 v[F / 2];
    __device__ __forceinline__ void set(const float (&x)[F]) { pack2<F>(x, v); }
    __device__ __forceinline__ static float dot(const PhiBuf& x, const PhiBuf& y) { return dot2<F / 2>(x.v, y.v); }
};
template <int A, int F, bool PK> struct WBuf {
    float w[A][F];
    __device__ __forceinline__ float get(int b, int f) const { return w[b][f]; }
    __device__ __forceinline__ void put(int b, int f, float x) { w[b][f] = x; }
    __device__ __forceinline__ void q(const PhiBuf<F, PK>& phi, float (&out)[A]) const { q_from_reg<A, F>(w, phi.v, out); }
    // W[:,b] += sb[b] * phi for every column (sb is zero off the action taken: no divergent select of the column)
    __device__ __forceinline__ void axpy(const float (&sb)[A], const PhiBuf<F, PK>& phi) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) w[b][f] = fmaf(sb[b], phi.v[f], w[b][f]);
    }
    // this += scale * other   (Handler<ScaledGradientUpdate>, fa/linear.rs:184-196)
    __device__ __forceinline__ void axpy_buf(float scale, const WBuf& o) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) w[b][f] = fmaf(scale, o.w[b][f], w[b][f]);
    }
    // trace decay + gradient: z[b] = fma(rate, z[b], ind[b] * phi), ind in {0, 1} (the product is exact; a multiply
    // instead of a per-element select: v_cndmask issues at half the FMA rate at one wave per SIMD)
    __device__ __forceinline__ void decay_add(float rate, const float (&ind)[A], const PhiBuf<F, PK>& phi) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) w[b][f] = fmaf(rate, w[b][f], ind[b] * phi.v[f]);
    }
    __device__ __forceinline__ void clip(float lo, float hi) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) w[b][f] = __builtin_amdgcn_fmed3f(w[b][f], lo, hi);      // one v_med3_f32, not min + max + canonicalise
    }
};
template <int A, int F> struct WBuf<A, F, true> {
    f2 w[A][F / 2];
    __device__ __forceinline__ float get(int b, int f) const { return (f & 1) ? w[b][f >> 1].y : w[b][f >> 1].x; }
    __device__ __forceinline__ void put(int b, int f, float x) { if (f & 1) w[b][f >> 1].y = x; else w[b][f >> 1].x = x; }
    __device__ __forceinline__ void q(const PhiBuf<F, true>& phi, float (&out)[A]) const { q_from_reg2<A, F / 2>(w, phi.v, out); }
    __device__ __forceinline__ void axpy(const float (&sb)[A], const PhiBuf<F, true>& phi) {
#pragma unroll
        for (int b = 0; b < A; ++b) {
            const f2 s2 = {sb[b], sb[b]};
#pragma unroll
            for (int j = 0; j < F / 2; ++j) w[b][j] = __builtin_elementwise_fma(s2, phi.v[j], w[b][j]);
        }
    }
    __device__ __forceinline__ void axpy_buf(float scale, const WBuf& o) {
        const f2 s2 = {scale, scale};
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int j = 0; j < F / 2; ++j) w[b][j] = __builtin_elementwise_fma(s2, o.w[b][j], w[b][j]);
    }
    __device__ __forceinline__ void decay_add(float rate, const float (&ind)[A], const PhiBuf<F, true>& phi) {
        const f2 r2 = {rate, rate};
#pragma unroll
        for (int b = 0; b < A; ++b) {
            const f2 i2 = {ind[b], ind[b]};
#pragma unroll
            for (int j = 0; j < F / 2; ++j) w[b][j] = __builtin_elementwise_fma(r2, w[b][j], i2 * phi.v[j]);
        }
    }
    __device__ __forceinline__ void clip(float lo, float hi) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int j = 0; j < F / 2; ++j) {
                w[b][j].x = __builtin_amdgcn_fmed3f(w[b][j].x, lo, hi);      // one v_med3_f32, not min + max + canonicalise
                w[b][j].y = __builtin_amdgcn_fmed3f(w[b][j].y, lo, hi);
            }
    }
};

// td_error with the agent's own policy: a wave-uniform branch keeps the common case (the agent shares the behaviour
// policy: compile-time kind, folded switch) exactly as it was
template <int A, int ALGO>
__device__ __forceinline__ float td_error_ap(const AlgoParams& alg, const PolicyParams& pol, const Common& c, float qsa,
                                             const float (&qn)[A], float r, bool term, const U4& xin, float& e) {
    if constexpr (ALGO == ALG_SARSA || ALGO == ALG_ESARSA) {
        if (!c.apol_same) return td_error<A>(alg, c.apol, qsa, qn, r, term, xin, e);
    }
    return td_error<A>(alg, pol, qsa, qn, r, term, xin, e);
}

// one dword of a wave-uniform row: descriptor built from the row pointer (SALU), lane offset in bytes
__device__ __forceinline__ float row_load(const char* rowp, uint32_t voff) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rowp, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
}
__device__ __forceinline__ void row_store(char* rowp, uint32_t voff, float v) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rowp, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, (int)voff, 0, 0);
}

// ---- selects on the ACTION as bit operations (round 3).  A v_cmp writes an SGPR pair and the v_cndmask behind it has to wait for
// it (gfx950: s_nop 1 between them; 8.4 cycles per select for a lone wave, profiles/r01_ubench_valu_issue.txt).  The action is a
// small integer, so "a == b" is available as an all-ones / all-zeros VECTOR mask from two integer instructions, and a select is
// one v_bfi_b32 / v_and_b32 on it: same values, bit for bit, no SGPR round trip.  The masks are made opaque to the optimiser, which
// would otherwise recognise sext(icmp) and fold the bit operations back into compare + select.
struct ActionMask {
    uint32_t m0, m1, m2;
    __device__ __forceinline__ explicit ActionMask(int a) {            // a in {0, 1, 2}
        m0 = sign_mask(a - 1);                                          // a == 0
        m2 = sign_mask(1 - a);                                          // a == 2
        m1 = ~(m0 | m2);
    }
    __device__ __forceinline__ uint32_t of(int b) const { return b == 0 ? m0 : (b == 1 ? m1 : m2); }
};
__device__ __forceinline__ float and_mask(uint32_t m, float x) {                 // m ? x : +0.0
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & m);
}

// q[a] by a compare/select chain.  Each compare sees its own opaque copy of the index: otherwise LLVM folds the
// chain into a dynamic extractelement, which is legalised through a private array that promote-alloca moves to LDS
// -- an exposed ds_write/ds_read round trip on the critical path of every step.
template <int A>
__device__ __forceinline__ float select_a(const float (&q)[A], int a) {
    float v = q[0];
#pragma unroll
    for (int i = 1; i < A; ++i) {
        int ai = a;
        asm("" : "+v"(ai));
        v = (ai == i) ? q[i] : v;
    }
    return v;
}

// Q(s,.) carried across loop iterations as named scalars: as an array it stays an alloca (promote-alloca then puts it
// in LDS and every step pays a ds_write/ds_read round trip on its critical path).
struct QCarry {
    float v0, v1, v2;
    template <int A> __device__ __forceinline__ void set(const float (&q)[A]) {
        v0 = q[0]; v1 = A > 1 ? q[A > 1 ? 1 : 0] : 0.0f; v2 = A > 2 ? q[A > 2 ? 2 : 0] : 0.0f;
    }
    // two selects; each compare sees its own opaque copy of the index (see select_a: LLVM would otherwise turn the chain
    // into a dynamic extractelement through LDS)
    __device__ __forceinline__ float at(int a) const {
        int a1 = a, a2 = a;
        asm("" : "+v"(a1));
        asm("" : "+v"(a2));
        const float x0 = v0, x1 = v1, x2 = v2;     // by value: ?: on two member lvalues is a select of ADDRESSES plus a load
        const float t = (a2 == 2) ? x2 : x0;
        return (a1 == 1) ? x1 : t;
    }
};

// argmaxima with its running maximum (utils.rs:6-21: the tolerance test comes first, the maximum is not raised by near-ties)
template <int A>
__device__ __forceinline__ uint32_t argmaxima_mask_max(const float (&q)[A], float& mx_out) {
    float mx = -FLT_MAX; uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const float d = fabsf(q[i] - mx);
        if (d < 1e-7f) mask |= (1u << i);
        else if (q[i] > mx) { mx = q[i]; mask = (1u << i); }
    }
    mx_out = mx;
    return mask;
}
// Function<(S, A)> of the four policies: greedy.rs:46-60, epsilon_greedy.rs:49-63, softmax.rs:84-92, random.rs:28-32
template <int A>
__device__ __forceinline__ float policy_eval_sa(const PolicyParams& pp, const float (&q)[A], int a) {
    if (pp.kind == POL_SOFTMAX) return select_a<A>(q, a);
    if (pp.kind == POL_RANDOM) return 1.0f / (float)A;
    float mx;
    const uint32_t mask = argmaxima_mask_max<A>(q, mx);
    const float pg = ((mask >> a) & 1u) ? 1.0f / (float)max(1, __popc(mask)) : 0.0f;
    if (pp.kind == POL_GREEDY) return pg;
    return pp.eps / (float)A + (1.0f - pp.eps) * pg;
}

// Enumerable::find_min: fold `if acc.1 < x {acc} else {(i,x)}` => the minimum, ties go to the LAST index   core.rs:86-94
template <int A>
__device__ __forceinline__ int find_min(const float (&q)[A], float& val) {
    int bi = 0; float bv = q[0];
#pragma unroll
    for (int i = 1; i < A; ++i) { if (!(bv < q[i])) { bi = i; bv = q[i]; } }
    val = bv;
    return bi;
}
// Enumerable::expected_value: zip(values, ps).fold(0.0, |acc, (x, p)| acc + x * p)                        core.rs:107-116
template <int A>
__device__ __forceinline__ float expected_value(const float (&q)[A], const float (&p)[A]) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) acc = acc + q[i] * p[i];
    return acc;
}

// ---------------------------------------------------------------------------------------
// The fused driver loop  (examples/q_learning.rs:40-52, order of operations SURVEY A.7)
//   per step:  t = env.transition(a)            lib.rs:436-446
//              agent.handle(&t)                 q_learning.rs:51-71 / sarsa.rs / expected_sarsa.rs
//              a = policy.sample(rng, s')       with the UPDATED weights
//              terminal or step cap -> fresh env + fresh policy.sample(s0)
// The reference projects phi 4x per step (s, s', s, s'); here phi(s) and Q(s,.) are
// carried over from the previous step (same W, same s => same values) and phi(s') is
// projected once.  n_steps batch-steps per launch; W, phi(s), Q(s,.) live in VGPRs.
//
// Shape of the loop body (one wave per SIMD at N = 65 536, so single-wave ILP is what counts):
//  * ALGO / POLICY are template parameters: no uniform branches splitting the body;
//  * a TERMINAL transition needs no Q(s') (delta = r - Q(s,a)), so the projection slot is
//    used for s0 instead: terminal resets cost nothing extra and do not diverge; only a
//    step-cap truncation (needs both Q(s') and Q(s0)) takes the divergent slow path;
//  * the loop is unrolled by two with ping-pong phi buffers (no phi(s) <- phi(s') moves).
// All A columns are written once at the end of the launch (one batch-step per launch is k_step_reg / k_step_reg_lm).
// ---------------------------------------------------------------------------------------

// Registers: __launch_bounds__(kBlock, 2) = at most 256 per lane, so a launch of more than 1024 waves (> 65 536 learners) runs
// TWO waves per SIMD: a lone wave issues one VALU instruction per ~3.4 cycles (packed fma 5.2, v_mad_u64 8, v_cndmask 8.4), two
// co-resident waves one per 2.7 / 4.8 / 5.9 / 4.3 (profiles/r01_ubench_valu_issue.txt).  The loop itself needs ~225 registers;
// what had pushed the first version to 460 (one wave per SIMD at EVERY size) was its prologue / epilogue: per-element 64-bit
// addresses, CSE'd between the loads and the stores and kept alive across the loop.
// ESCHED: the per-learner epsilon schedule (Common::eps) -- an instantiation of its own, so that the schedule-free loop (the bench's)
// keeps its threshold in a scalar register and not one instruction more.
template <int DOMAIN, int ORDER, int ALGO, int POLICY, bool ESCHED = false>
__global__ __launch_bounds__(kBlock, 2) void k_train_reg(Common c, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = c.n_envs;
    // W rows through a per-row buffer descriptor (wave-uniform base in SGPRs) + a 32-bit lane offset: one buffer_load /
    // buffer_store per element, no per-element 64-bit VALU address arithmetic and no address registers kept alive from the
    // loads to the stores (the first version spent ~900 instructions of its entry block on addresses and, limited to 256
    // registers, spilled all of them across the loop)
    char* __restrict__ const wblk = reinterpret_cast<char*>(c.W + (int64_t)blockIdx.x * blockDim.x * c.w_ls);
    const uint32_t wlane = threadIdx.x * (uint32_t)c.w_ls * 4u;          // byte offset of this lane inside a row, 32 bits
    const int64_t wrow = c.w_stride * 4;                                 // bytes between (action, feature) rows

    uint32_t n_ep = 0, n_trunc = 0, sum_len = 0;      // per launch and learner: <= n_steps (+ the carried episode length), 32 bits
    double sum_abs = 0.0, sum_r = 0.0;

    if (i < N) {
        PolicyParams pol = c.pol; pol.kind = POLICY;
        if constexpr (ESCHED) learner_eps_load(c, i, pol);
        AlgoParams alg = c.alg; alg.kind = ALGO;
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        float s[D];
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
        int a = c.action[i];
        uint32_t ep = c.ep_step[i];
        const uint32_t ep_start = ep;
        constexpr bool PK = (RSRL_PK != 0) && (F % 4 == 0);
        using Phi = PhiBuf<F, PK>;
        WBuf<A, F, PK> w;
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) w.put(b, f, row_load(wblk + (int64_t)(b * F + f) * wrow, wlane));

        static_assert(A <= 3, "QCarry holds up to 3 actions");
        Phi phi_a, phi_b;
        QCarry q_s;
        { float ph[F]; Bas::project(s, ph); phi_a.set(ph); }
        {
            float q0[A];
            if (c.q_valid) {             // Q(s,.) carried from the previous launch (bit-identical to not having stopped)
#pragma unroll
                for (int b = 0; b < A; ++b) q0[b] = c.qcache[(int64_t)b * N + i];
            } else {
                w.q(phi_a, q0);
            }
            q_s.set<A>(q0);
        }
        typename Dom::Pre pre_s = Dom::pre(s);
        float facc_abs = 0.0f, facc_r = 0.0f;       // fp32 partial sums, flushed to f64 every launch
        // every load of the prologue retires HERE: otherwise the compiler, which sees the weight loads pending on the loop's entry edge
        // only, places their waits at the first uses INSIDE the loop -- ~19 s_waitcnt per pair of steps that wait for nothing
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)

        // x = this batch-step's behaviour-policy draw, xin = the agent's own (SARSA): halves of Philox blocks shared by two steps
        auto one_step = [&](const Phi& phi_s, Phi& phi_n, const U4& x, const U4& xin) {
            // ---- Domain::transition
            float ns[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            const bool term = Dom::step(ns, a, r, pre_s);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            if (term) Dom::reset(ns);              // select, not a branch: phi/Q of s0 take the s' slot
            pre_s = Dom::pre(ns);                  // action-independent part of the NEXT transition, off the critical path
            float q_n[A];
            { float ph[F]; Bas::project(ns, ph); phi_n.set(ph); }
            w.q(phi_n, q_n);
            // ---- handle: delta with the PRE-update weights
            static_assert(A <= 3, "ActionMask covers three actions");
            const ActionMask am(a);
            const float qsa = A > 2 ? bitsel(am.m2, q_s.v2, bitsel(am.m1, q_s.v1, q_s.v0)) : bitsel(am.m1, q_s.v1, q_s.v0);
            const float q_s_all[3] = {q_s.v0, q_s.v1, q_s.v2};
            float e;
            float delta;
            if constexpr (ALGO == ALG_PAL) {
                float qs_arr[A];
#pragma unroll
                for (int b = 0; b < A; ++b) qs_arr[b] = q_s_all[b];
                delta = td_error_pal<A>(alg, qs_arr, q_n, a, r, term, e);
            } else {
                delta = td_error_ap<A, ALGO>(alg, pol, c, qsa, q_n, r, term, xin, e);
            }
            // ---- Handler<StateActionUpdate>: W[:,a] += lr * e * phi(s)     fa/linear.rs:379-391
            const float scale = alg.lr * e;
            {
                float sb[A];
#pragma unroll
                for (int b = 0; b < A; ++b) sb[b] = and_mask(am.of(b), scale);
                w.axpy(sb, phi_s);
            }
            // ---- policy.sample with the UPDATED weights (at s', or at s0 after a terminal transition)
            {   // W changed by a rank-1 term in column a only: Q_post[a] = Q_pre[a] + scale * <phi(s), phi(s')>
                const float dot = Phi::dot(phi_s, phi_n);
#pragma unroll
                for (int b = 0; b < A; ++b) q_n[b] = bitsel(am.of(b), fmaf(scale, dot, q_n[b]), q_n[b]);
            }
            if constexpr (ESCHED) learner_eps_step(c, term | trunc, pol);        // the episode's last handle is done: its end decays epsilon
            int na = policy_sample<A, true>(pol, q_n, x);               // (x is a half block: y == z)
            facc_abs += fabsf(delta);
            facc_r += r;
            n_ep += term ? 1u : 0u; ep = term ? 0u : ep;
            if (__builtin_expect(trunc, 0)) {      // step cap: Q(s') was needed above, now the new episode
                n_ep += 1; n_trunc += 1; ep = 0;
                Dom::reset(ns);
                pre_s = Dom::pre(ns);
                { float ph[F]; Bas::project(ns, ph); phi_n.set(ph); }
                w.q(phi_n, q_n);
                na = policy_sample<A, true>(pol, q_n, x);    // the step's one behaviour sample: the same draw (BLK_RESET == BLK_STEP)
            }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
            q_s.set<A>(q_n);
            a = na;
        };

        constexpr bool INNER = ALGO == ALG_SARSA;                    // the only agent that draws for itself on this path
        auto single = [&](uint64_t t) {                               // a step outside a pair: its half of the block
            one_step(phi_a, phi_b, draw(c.seed, gid, t, BLK_STEP), INNER ? draw(c.seed, gid, t, BLK_INNER) : U4{0, 0, 0, 0});
        };
        int k = 0;
        if ((t0 & 1u) && n_steps > 0) {                               // pairs start at even batch-steps
            single(t0);
            const Phi tmp = phi_a; phi_a = phi_b; phi_b = tmp;        // phi(s) back into the first buffer (once per launch)
            k = 1;
        }
        for (; k + 1 < n_steps; k += 2) {
            const uint64_t th = (t0 + (uint64_t)k) >> 1;              // ONE Philox block per stream for the two steps
            const U4 p = draw_block(c.seed, gid, th, BLK_STEP);
            const U4 pin = INNER ? draw_block(c.seed, gid, th, BLK_INNER) : U4{0, 0, 0, 0};
            one_step(phi_a, phi_b, half_block(p, false), half_block(pin, false));
            one_step(phi_b, phi_a, half_block(p, true), half_block(pin, true));
        }
        if (k < n_steps) single(t0 + (uint64_t)k);
        sum_abs = (double)facc_abs; sum_r = (double)facc_r;
        // lengths of the episodes that ended in this launch: every step taken, plus what the first episode had before the launch,
        // minus what the open one has now
        sum_len = (uint32_t)n_steps + ep_start - ep;

#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
        if constexpr (ESCHED) c.eps[i] = pol.eps;
        c.qcache[i] = q_s.v0;
        if constexpr (A > 1) c.qcache[N + i] = q_s.v1;
        if constexpr (A > 2) c.qcache[2 * N + i] = q_s.v2;
        // the row addresses are computed AGAIN here (the base laundered through an empty asm statement): kept from the loads they are
        // 216 scalar registers that live across the whole loop -- spilled into VGPR lanes (370 v_writelane before the loop, 370
        // v_readlane after it) and, at 256 VGPRs, pushing vector registers into scratch
        char* wblk_e = wblk;
        int64_t wrow_e = wrow;
        asm volatile("" : "+s"(wblk_e), "+s"(wrow_e));
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) row_store(wblk_e + (int64_t)(b * F + f) * wrow_e, wlane, w.get(b, f));
    }

    // per-launch statistics: block-level reduction into this block's own slot (no atomics: a slot
    // has exactly one writer per launch, and launches are ordered on the ctx's stream)
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// ---------------------------------------------------------------------------------------
// The single-step streaming kernel (K = 1): one launch = one batch-step, every learner's W
// streamed from memory once -- the 608 B/env-step formulation:
//   read  state 8 + action 4 + ep_step 4 + W 3*36*4 = 448 B,  write state 8 + action 4 + ep_step 4 + W[:,a] 144 B.
// Compact on purpose: a cold launch is instruction-fetch bound, so W goes through ONE buffer
// descriptor (SGPR row offset + 4*lane, no per-load 64-bit VALU address math), all 108 loads are
// issued before any arithmetic, and only the touched column is kept/selected/stored.
// Arithmetic is bit-identical to k_train_reg (same helpers, same op order).
// Precondition (checked by the host): A*F*N*4 < 2^32.
// ---------------------------------------------------------------------------------------
template <int DOMAIN, int ORDER, int ALGO, int POLICY>
__global__ __launch_bounds__(kBlock) void k_step_reg(Common c, uint64_t t, DevStats* __restrict__ stats, const uint64_t* __restrict__ t_dev) {
    if (t_dev) t += *t_dev;            // graph replay: the batch-step counter lives on the device, t is the node's offset
    if (c.dyn) { c.pol = c.dyn->pol; c.apol = c.dyn->apol; }
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F;
    const int64_t N = c.n_envs;
    const int64_t base = (int64_t)blockIdx.x * kBlock;                 // wave-uniform
    const int64_t i = base + threadIdx.x;
    const uint32_t row_bytes = (uint32_t)N * 4u;
    const uint32_t total_bytes = (uint32_t)(A * F) * row_bytes - (uint32_t)base * 4u;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(c.W + base), 0, (int)total_bytes, 0x00020000);
    const int voff = (int)threadIdx.x * 4;

    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;

    // Load order matters: vmcnt retires in issue order, so whatever the first arithmetic needs (state, action,
    // the carried Q) is requested BEFORE the 108 weight loads -- the transition, both projections and the
    // Philox rounds then run underneath the weight stream instead of behind it.
    const int64_t il = i < N ? i : N - 1;                              // clamped: the loads are unconditional
    float s[D], ns[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + il];
    const int a = c.action[il];
    uint32_t ep = c.ep_step[il] + 1;
    float qc0 = 0.0f, qc1 = 0.0f, qc2 = 0.0f;
    static_assert(A <= 3, "carried-Q registers are laid out for A <= 3");
    if (c.q_valid) {
        qc0 = c.qcache[il];
        qc1 = c.qcache[N + il];
        if constexpr (A > 2) qc2 = c.qcache[2 * N + il];
    }
    // all weight loads in flight (out-of-range lanes of the last block read in-bounds garbage or 0)
    float wv[A][F];
#pragma unroll
    for (int b = 0; b < A; ++b)
#pragma unroll
        for (int f = 0; f < F; ++f)
            wv[b][f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)((uint32_t)(b * F + f) * row_bytes), 0));

    if (i < N) {
        PolicyParams pol = c.pol; pol.kind = POLICY;
        AlgoParams alg = c.alg; alg.kind = ALGO;
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
#pragma unroll
        for (int d = 0; d < D; ++d) ns[d] = s[d];
        // ---- Domain::transition
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
        // ---- Q(s,a): carried from the previous launch, or recomputed when the cache is stale
        constexpr int P = RSRL_DOT_SPLIT;
        float qs_arr[A];
        if (c.q_valid) {
            qs_arr[0] = qc0;
            if constexpr (A > 1) qs_arr[1] = qc1;
            if constexpr (A > 2) qs_arr[2] = qc2;
        } else {
            q_from_reg<A, F>(wv, phi_s, qs_arr);
        }
        const float qsa = (a == 0) ? qs_arr[0] : ((a == 1) ? qs_arr[A > 1 ? 1 : 0] : qs_arr[A > 2 ? 2 : 0]);
        q_from_reg<A, F>(wv, phi_n, q_n);                              // Q(s',.) with the PRE-update weights
        float e, delta;
        if constexpr (ALGO == ALG_PAL) delta = td_error_pal<A>(alg, qs_arr, q_n, a, r, term, e);
        else delta = td_error_ap<A, ALGO>(alg, pol, c, qsa, q_n, r, term, xin, e);
        // ---- W[:,a] += lr * e * phi(s); every (action, feature) row goes back as a FULL line (the untouched
        //      columns are rewritten unchanged: the lane-dependent column would dirty all three lines anyway)
        const float scale = alg.lr * e;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float v = wv[0][f];
#pragma unroll
            for (int b = 1; b < A; ++b) v = (a == b) ? wv[b][f] : v;
            v = fmaf(scale, phi_s[f], v);
#pragma unroll
            for (int b = 0; b < A; ++b) {
                wv[b][f] = (a == b) ? v : wv[b][f];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, wv[b][f]), rs, voff, (int)((uint32_t)(b * F + f) * row_bytes), 0);
            }
        }
        // ---- Q(s',.) with the UPDATED weights: only column a changed
        {
            float dacc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) dacc[p] = 0.0f;
#pragma unroll
            for (int f = 0; f < F; ++f) dacc[f % P] = fmaf(phi_s[f], phi_n[f], dacc[f % P]);
            const float dot = combine_partials<P>(dacc);
#pragma unroll
            for (int b = 0; b < A; ++b) q_n[b] = (a == b) ? fmaf(scale, dot, q_n[b]) : q_n[b];
        }
        int na = policy_sample<A>(pol, q_n, x);
        sum_abs = (double)fabsf(delta); sum_r = (double)r;
        if (term) { n_ep = 1; sum_len = ep; ep = 0; }
        if (trunc) {                               // step cap: new episode needs Q(s0) with the updated W
            n_ep = 1; n_trunc = 1; sum_len = ep; ep = 0;
            Dom::reset(ns);
            Bas::project(ns, phi_n);
            q_from_reg<A, F>(wv, phi_n, q_n);
            const U4 xr = draw(c.seed, gid, t, BLK_RESET);
            na = policy_sample<A>(pol, q_n, xr);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = ns[d];
        c.action[i] = na;
        c.ep_step[i] = ep;
#pragma unroll
        for (int b = 0; b < A; ++b) c.qcache[(int64_t)b * N + i] = q_n[b];
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// ---------------------------------------------------------------------------------------
// The single-step streaming kernel on the LEARNER-MAJOR layout W[N][A][F] (what a ctx created with steps_per_launch = 1
// uses when A*F is a multiple of 4).  With the feature-major layout every (action, feature) row is a full line per wave and
// the lane-dependent column dirties all A lines of every feature: 432 B written per learner for 144 B changed.  Here a
// learner's weights are contiguous:
//   read : the wave's 64 learners = one contiguous 64*A*F*4 B image, coalesced 16 B per lane, transposed through LDS
//          (ds_write_b128 linear, ds_read_b128 at lane*A*F*4 + 16k: with A*F = 108 dwords the 16 lanes of a pass hit
//          disjoint banks);
//   write: ONLY the touched column, F*4 contiguous bytes per learner (16 B stores); the column's old values are re-read
//          from the LDS image at a lane-dependent address, so no select chains.
// => 432 B read + 144 B written per learner-step, the 608 B/env-step accounting of SURVEY 8(d), instead of 432 + 432.
// Arithmetic is bit-identical to k_step_reg / k_train_reg (same helpers, same op order).
// Precondition (host): per-env weights, (A*F) % 4 == 0, F % 4 == 0, A*F*N*4 < 2^32.
// ---------------------------------------------------------------------------------------
template <int DOMAIN, int ORDER, int ALGO, int POLICY>
__global__ __launch_bounds__(kBlock) void k_step_reg_lm(Common c, uint64_t t, DevStats* __restrict__ stats, const uint64_t* __restrict__ t_dev) {
    if (t_dev) t += *t_dev;
    if (c.dyn) { c.pol = c.dyn->pol; c.apol = c.dyn->apol; }
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F, AF = A * F;
    static_assert(AF % 4 == 0 && F % 4 == 0, "16-byte rows");
    constexpr int AF4 = AF / 4, F4 = F / 4;
    __shared__ __attribute__((aligned(16))) float lds[kBlock * AF];
    const int64_t N = c.n_envs;
    const int lane = (int)(threadIdx.x & 63), wv_id = (int)(threadIdx.x >> 6);
    const int64_t wbase = (int64_t)blockIdx.x * kBlock + wv_id * 64;    // first learner of this wave
    const int64_t i = wbase + lane;
    float* __restrict__ img = lds + wv_id * 64 * AF;                    // this wave's 64 x A x F image
    // buffer descriptor over the wave's image: loads beyond the end of W return 0, stores are dropped (last, partial wave)
    const int64_t remain = N - wbase;
    const uint32_t img_bytes = (uint32_t)((remain < 64 ? (remain < 0 ? 0 : remain) : 64) * AF * 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(c.W + wbase * AF), 0, (int)img_bytes, 0x00020000);

    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;

    // the small loads the first arithmetic needs go out BEFORE the weight stream (vmcnt retires in issue order)
    const int64_t il = i < N ? i : N - 1;
    float s[D], ns[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + il];
    const int a = c.action[il];
    uint32_t ep = c.ep_step[il] + 1;
    float qc0 = 0.0f, qc1 = 0.0f, qc2 = 0.0f;
    static_assert(A <= 3, "carried-Q registers are laid out for A <= 3");
    if (c.q_valid) {
        qc0 = c.qcache[il];
        qc1 = c.qcache[N + il];
        if constexpr (A > 2) qc2 = c.qcache[2 * N + il];
    }
    // the image: AF4 coalesced 16-B loads per lane, all in flight ...
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef int i4 __attribute__((ext_vector_type(4)));
    // straight into the wave's LDS image (buffer_load_dwordx4 ... lds: wave-uniform LDS base + lane * 16): no staging registers,
    // no ds_write pass; beyond the end of W the descriptor returns zeros
#pragma unroll
    for (int m = 0; m < AF4; ++m)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(img + 64 * 4 * m), 16, lane * 16, 64 * 16 * m, 0, 0);
    // where this lane will store in the write-back of the touched columns (see below): piece (g & 3) of sector t of learner j's
    // column -- known from the actions alone, so the lane exchange runs here, under the loads
    constexpr int NSEC = (48 + F * 4 + 63) / 64;                        // sectors a 16-byte-aligned column can touch
    int wb_off[4 * NSEC];
    {
        const int sec = i < N ? (((lane * AF + a * F) * 4) & ~63) : 0x40000000;       // learner beyond N: outside the descriptor's range
#pragma unroll
        for (int p = 0; p < 4 * NSEC; ++p) {
            const int g = p * 64 + lane, sidx = g >> 2, j = sidx / NSEC, t = sidx - j * NSEC;
            wb_off[p] = __builtin_amdgcn_ds_bpermute(j * 4, sec) + 64 * t + 16 * (g & 3);
        }
    }

    // ... and everything that needs only the state runs underneath them: the transition, both projections, the draws
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

    // transpose through LDS: linear 16-B writes, then each lane reads its own learner's A*F weights
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

    // every lane computes (a lane beyond N repeats learner N-1 and stores nothing): no divergence around the wave-level write-back
    {
        constexpr int P = RSRL_DOT_SPLIT;
        float qs_arr[A];
        if (c.q_valid) {
            qs_arr[0] = qc0;
            if constexpr (A > 1) qs_arr[1] = qc1;
            if constexpr (A > 2) qs_arr[2] = qc2;
        } else {
            q_from_reg<A, F>(wv, phi_s, qs_arr);
        }
        const float qsa = (a == 0) ? qs_arr[0] : ((a == 1) ? qs_arr[A > 1 ? 1 : 0] : qs_arr[A > 2 ? 2 : 0]);
        q_from_reg<A, F>(wv, phi_n, q_n);                              // Q(s',.) with the PRE-update weights
        float e, delta;
        if constexpr (ALGO == ALG_PAL) delta = td_error_pal<A>(alg, qs_arr, q_n, a, r, term, e);
        else delta = td_error_ap<A, ALGO>(alg, pol, c, qsa, q_n, r, term, xin, e);
        // ---- W[:,a] += lr * e * phi(s): old column from the LDS image (lane-dependent address), new column to memory
        const float scale = alg.lr * e;
        float vcol[F];
        float* __restrict__ colp = img + lane * AF + a * F;
        [[maybe_unused]] const int col_off = (lane * AF + a * F) * 4;
#pragma unroll
        for (int k = 0; k < F4; ++k) {
            const f4 o = *reinterpret_cast<const f4*>(colp + 4 * k);
            f4 v;
            v.x = fmaf(scale, phi_s[4 * k], o.x); v.y = fmaf(scale, phi_s[4 * k + 1], o.y);
            v.z = fmaf(scale, phi_s[4 * k + 2], o.z); v.w = fmaf(scale, phi_s[4 * k + 3], o.w);
            vcol[4 * k] = v.x; vcol[4 * k + 1] = v.y; vcol[4 * k + 2] = v.z; vcol[4 * k + 3] = v.w;
            *reinterpret_cast<f4*>(colp + 4 * k) = v;                     // merged into the wave's image, written back below
        }
        // The touched column goes back as WHOLE 64-byte sectors: F*4 = 144 bytes at a 16-byte-aligned offset dirty three sectors, and
        // what the memory side is slow at is a store instruction that scatters 64 separate 16-byte pieces (scripts/ubench/
        // stream_pattern.hip: read 432 + write 144 per learner, no arithmetic, 6.2 us per launch at 65 536 learners; 4.7 us with the
        // same columns written as whole sectors, four lanes per sector).  So the lanes' updates are merged into the wave's LDS image,
        // and the wave then writes its 64 x 3 dirty sectors with FOUR CONSECUTIVE LANES PER SECTOR (the sector offsets come from the
        // owning lanes by ds_bpermute): every store instruction writes 16 whole sectors.  Measured, us per launch: 9.05 -> 7.94 at
        // 65 536 learners (0.53 -> 0.60 of 8 TB/s on the 608 B/env-step accounting), 21.3 -> 19.6 at 131 072, 37.9 -> 36.0 at 262 144.
        // Each lane writing its own three sectors (variant 1: still 64 scattered pieces per instruction) was SLOWER than the direct
        // stores: 10.3 us.  Two learners whose sectors overlap store the same bytes (both read the merged image); the image is a
        // multiple of 64 bytes long, so no sector is shared between waves; sectors of learners beyond N fall outside the descriptor.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            static_assert((64 * AF * 4) % 64 == 0, "no sector is shared between two waves' images");
            // four consecutive lanes store one sector: every store instruction writes 16 whole sectors
#pragma unroll
            for (int p = 0; p < 4 * NSEC; ++p) {
                const f4 v = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(img) + (wb_off[p] & 0xffff));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, wb_off[p], 0, 0);
            }
        }
        // ---- Q(s',.) with the UPDATED weights: only column a changed
        {
            float dacc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) dacc[p] = 0.0f;
#pragma unroll
            for (int f = 0; f < F; ++f) dacc[f % P] = fmaf(phi_s[f], phi_n[f], dacc[f % P]);
            const float dot = combine_partials<P>(dacc);
#pragma unroll
            for (int b = 0; b < A; ++b) q_n[b] = (a == b) ? fmaf(scale, dot, q_n[b]) : q_n[b];
        }
        int na = policy_sample<A>(pol, q_n, x);
        sum_abs = (double)fabsf(delta); sum_r = (double)r;
        if (term) { n_ep = 1; sum_len = ep; ep = 0; }
        if (trunc) {                               // step cap: new episode needs Q(s0) with the updated W
            n_ep = 1; n_trunc = 1; sum_len = ep; ep = 0;
            Dom::reset(ns);
            Bas::project(ns, phi_n);
            q_from_reg<A, F>(wv, phi_n, q_n);      // untouched columns: old == new
            float one[1][F], qa[1];
#pragma unroll
            for (int f = 0; f < F; ++f) one[0][f] = vcol[f];
            q_from_reg<1, F>(one, phi_n, qa);      // the touched column, same dot-product order
#pragma unroll
            for (int b = 0; b < A; ++b) q_n[b] = (a == b) ? qa[0] : q_n[b];
            const U4 xr = draw(c.seed, gid, t, BLK_RESET);
            na = policy_sample<A>(pol, q_n, xr);
        }
        if (i < N) {
#pragma unroll
            for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = ns[d];
            c.action[i] = na;
            c.ep_step[i] = ep;
#pragma unroll
            for (int b = 0; b < A; ++b) c.qcache[(int64_t)b * N + i] = q_n[b];
        } else {
            n_ep = 0; n_trunc = 0; sum_len = 0; sum_abs = 0.0; sum_r = 0.0;
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// ---------------------------------------------------------------------------------------
// Trait-granular kernels (drop-in use, fine-grained parity).  W is read from memory.
// ---------------------------------------------------------------------------------------

// Domain::transition on the ctx's envs      rsrl_domains/src/lib.rs:436-446
template <int DOMAIN>
__global__ __launch_bounds__(kBlock) void k_domain_step(Common c, const int32_t* __restrict__ actions,
                                                        float* __restrict__ from_out, float* __restrict__ next_out,
                                                        float* __restrict__ rew_out, uint8_t* __restrict__ term_out) {
    using Dom = Domain<DOMAIN>;
    constexpr int D = Dom::D;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float s[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        s[d] = c.state[(int64_t)d * N + i];
        if (from_out) from_out[(int64_t)d * N + i] = s[d];
    }
    const int a = actions ? actions[i] : c.action[i];
    float r;
    const bool term = Dom::step(s, a, r);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        c.state[(int64_t)d * N + i] = s[d];
        if (next_out) next_out[(int64_t)d * N + i] = s[d];
    }
    if (rew_out) rew_out[i] = r;
    if (term_out) term_out[i] = term ? 1 : 0;
}
template <int DOMAIN>
__global__ __launch_bounds__(kBlock) void k_domain_reset(Common c, const uint8_t* __restrict__ mask) {
    using Dom = Domain<DOMAIN>;
    constexpr int D = Dom::D;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (mask && !mask[i]) return;
    float s[D]; Dom::reset(s);
#pragma unroll
    for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
    c.ep_step[i] = 0;
}

}  // namespace rsrl
