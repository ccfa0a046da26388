// kernels_td_tile.hpp -- prediction (TD / TDLambda, rsrl/src/prediction/td/td.rs:31-59, td_lambda.rs:41-78) on TILE CODING,
// one weight vector f32[F] per learner (ScalarLFA: Parameterised::weights_dim = (F, 1), fa/linear.rs:201-251).
//   V(s) = the sum of the T active weights, in tiling order;  grad V = 1.0 at the T active entries
//   TD       : td = r + gamma*V(s') - V(s) (terminal: r - V(s));  w[active] += lr * td
//   TDLambda : z <- rule(rate*z + grad) on EVERY entry first (traces.rs:188-240), then w += td * z -- the step IS the TD error
//              (td_lambda.rs:59-62) --, a terminal transition resets the trace.
// One block per learner, as kernels_lambda_tile.hpp: the scalar work redundantly in every thread, TDLambda's dense trace table
// swept once per step with 16-byte accesses (the T active entries set aside and redone with grad = 1), TD touching its T entries
// only.  Behaviour policy Random (prediction agents have no Q function).  Bit-identical to orc_handle_td on tile coding.
#pragma once

#include "kernels_td.hpp"

namespace rsrl {

template <int DOMAIN, int T>
__device__ __forceinline__ float v_tile(const float* __restrict__ wl, const typename TileModel<DOMAIN, T>::Feat& ft) {
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t) acc = acc + wl[ft.idx[t]];
    return acc;
}

// from == nullptr: the driver loop (n_steps batch-steps of learner blockIdx.x); otherwise Handler::handle on one caller-supplied
// transition per learner
template <int DOMAIN, int T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_td_tile(Common c, BasisGeom g, TdParams tp, int lambda, uint64_t t0, int n_steps, DevStats* __restrict__ stats,
                                                   const float* __restrict__ from, const float* __restrict__ rew, const float* __restrict__ to,
                                                   const uint8_t* __restrict__ termf, int64_t Mn, float* __restrict__ td_out) {
    using M = TileModel<DOMAIN, T>;
    constexpr int D = M::D, A = M::A;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t i = blockIdx.x;
    const int64_t N = c.n_envs;
    const int tid = (int)threadIdx.x;
    const int F = g.F;
    float* __restrict__ wl = c.W + i * (int64_t)F;
    float* __restrict__ zl = lambda ? tp.Z + i * (int64_t)F : nullptr;
    PolicyParams pol = c.pol; pol.kind = POL_RANDOM;
    const uint32_t gid = (uint32_t)(c.env_offset + i);
    const uint32_t cap = c.max_episode_steps;
    const bool driver = from == nullptr;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    float s[D];
    int a = 0; uint32_t ep = 0;
    if (driver) {
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
        a = c.action[i]; ep = c.ep_step[i];
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = from[(int64_t)d * Mn + i];
    }
    typename M::Feat fs, fn;
    M::features(s, g, fs);
    const float q0[A] = {};
    for (int k = 0; k < (driver ? n_steps : 1); ++k) {
        const uint64_t t = t0 + (uint64_t)k;
        float ns[D], r;
        bool term, trunc = false;
        if (driver) {
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            term = M::Dom::step(ns, a, r);
            ep += 1;
            trunc = !term && cap > 0 && ep >= cap;
            if (term) M::Dom::reset(ns);
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = to[(int64_t)d * Mn + i];
            r = rew[i]; term = termf[i] != 0;
        }
        M::features(ns, g, fn);
        const float v_s = v_tile<DOMAIN, T>(wl, fs), v_n = v_tile<DOMAIN, T>(wl, fn);
        const float td = term ? (r - v_s) : (r + c.alg.gamma * v_n - v_s);
        // ---- the T active entries: old values set aside
        float zo = 0.0f, wo = 0.0f; int ea = 0;
        if (tid < T) {
            ea = fs.idx[0];
#pragma unroll
            for (int tt = 1; tt < T; ++tt) ea = (tid == tt) ? fs.idx[tt] : ea;
            wo = wl[ea];
            if (lambda) zo = zl[ea];
        }
        __syncthreads();                                        // every gather of w above precedes every store below
        if (lambda) {
            for (int j = tid * 4; j < F; j += BLOCK * 4) {
                const f4 z4 = *reinterpret_cast<const f4*>(zl + j);
                f4 w4 = *reinterpret_cast<const f4*>(wl + j);
                f4 zz;
                zz.x = trace_merge(tp.trace, tp.rate, z4.x, 0.0f); zz.y = trace_merge(tp.trace, tp.rate, z4.y, 0.0f);
                zz.z = trace_merge(tp.trace, tp.rate, z4.z, 0.0f); zz.w = trace_merge(tp.trace, tp.rate, z4.w, 0.0f);
                w4.x = fmaf(td, zz.x, w4.x); w4.y = fmaf(td, zz.y, w4.y); w4.z = fmaf(td, zz.z, w4.z); w4.w = fmaf(td, zz.w, w4.w);
                *reinterpret_cast<f4*>(wl + j) = w4;
                *reinterpret_cast<f4*>(zl + j) = term ? f4{0.0f, 0.0f, 0.0f, 0.0f} : zz;
            }
            __syncthreads();
        }
        if (tid < T) {
            if (lambda) {
                const float zz = trace_merge(tp.trace, tp.rate, zo, 1.0f);
                wl[ea] = fmaf(td, zz, wo);
                zl[ea] = term ? 0.0f : zz;
            } else {
                wl[ea] = fmaf(c.alg.lr * td, 1.0f, wo);         // w += lr * td * phi, phi = 1 at the active entries
            }
        }
        __syncthreads();
        if (!driver) { if (tid == 0 && td_out) td_out[i] = td; break; }
        const U4 x = draw(c.seed, gid, t, BLK_STEP);
        int na = policy_sample<A>(pol, q0, x);
        sum_abs += (double)fabsf(td); sum_r += (double)r;
        if (term) { n_ep += 1; sum_len += ep; ep = 0; }
        if (trunc) {
            n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
            M::Dom::reset(ns);
            M::features(ns, g, fn);
            const U4 xr = draw(c.seed, gid, t, BLK_RESET);
            na = policy_sample<A>(pol, q0, xr);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = ns[d];
        fs = fn;
        a = na;
    }
    if (driver && tid == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
    }
    if (stats) {
        const bool me = tid == 0;
        block_stats_accumulate(stats, me ? n_ep : 0ull, me ? n_trunc : 0ull, me ? sum_len : 0ull, me ? sum_abs : 0.0, me ? sum_r : 0.0);
    }
}

// Function<(S,)>::evaluate of the ScalarLFA on tile coding: V(s_i) with learner i's weights
template <int DOMAIN, int T>
__global__ __launch_bounds__(kBlock) void k_v_tile(Common c, BasisGeom g, const float* __restrict__ states, int64_t Mn, float* __restrict__ out) {
    using M = TileModel<DOMAIN, T>;
    constexpr int D = M::D;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    float s[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = states[(int64_t)d * Mn + i];
    typename M::Feat ft;
    M::features(s, g, ft);
    out[i] = v_tile<DOMAIN, T>(c.W + i * (int64_t)g.F, ft);
}

}  // namespace rsrl
