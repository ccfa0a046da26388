// kernels_gq.hpp -- GreedyGQ on the register family:
//   GreedyGQ::handle   rsrl/src/control/td/greedy_gq.rs:73-141      driver rsrl/examples/greedy_gq.rs:21-58
// Two approximators per learner: fa_q (weights W, SGD(lr)) and fa_td (weights V, SGD(lr_td)), both f32[A][F][N].
// Per transition, all against the PRE-update matrices:
//     qsa = <phi(s), W[:,a]>,  td_est = <phi(s), V[:,a]>,  (na, qmax) = find_max(Q(s',.))      (ties -> last, core.rs:96-105)
//     td_error = r - qsa                         (terminal)          | r + gamma*qmax - qsa    (otherwise)
//     W[:,a]  += lr * td_error * phi(s);   then (non-terminal)  W[:,na] += lr * (-gamma*td_est) * phi(s')
//     V[:,a]  += lr_td * (td_error - td_est) * phi(s)
// In the fused driver loop W AND V stay in registers for the whole launch (the layout of the eligibility-trace kernels,
// kernels_lambda.hpp); V lives in the ctx's auxiliary matrix (the one the lambda agents use for the trace).
#pragma once

#include "models.hpp"

namespace rsrl {

enum : int { ALG_GREEDY_GQ = 6 };

struct GqParams {
    float* V;          // fa_td weights [A][F][N]
    float lr_td;       // SGD rate of fa_td
};

template <int DOMAIN, int ORDER, int POLICY>
__global__ __launch_bounds__(kBlock) void k_train_gq(Common c, GqParams gp, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = c.n_envs;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
        PolicyParams pol = c.pol; pol.kind = POLICY;
        const float gamma = c.alg.gamma, lr = c.alg.lr;
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        float s[D];
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
        int a = c.action[i];
        uint32_t ep = c.ep_step[i];
        constexpr bool PK = (RSRL_PK != 0) && (F % 4 == 0);
        using Phi = PhiBuf<F, PK>;
        WBuf<A, F, PK> w, v;
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) {
                w.put(b, f, c.W[((int64_t)(b * F + f)) * N + i]);
                v.put(b, f, gp.V[((int64_t)(b * F + f)) * N + i]);
            }
        Phi phi_a, phi_b;
        float q_s[A];
        { float ph[F]; Bas::project(s, ph); phi_a.set(ph); }
        w.q(phi_a, q_s);
        float facc_abs = 0.0f, facc_r = 0.0f;

        auto one_step = [&](const Phi& phi_s, Phi& phi_n, uint64_t t) {
            float ns[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            const bool term = Dom::step(ns, a, r);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            if (term) Dom::reset(ns);                       // a terminal transition never reads Q(s',.): go straight to the restart state
            float q_n[A], e_s[A];
            { float ph[F]; Bas::project(ns, ph); phi_n.set(ph); }
            w.q(phi_n, q_n);
            v.q(phi_s, e_s);                                // all A columns, then select: no per-element column select
            const float qsa = select_a<A>(q_s, a);
            const float td_est = select_a<A>(e_s, a);
            float qmax;
            const int na_star = find_max<A>(q_n, qmax);
            const float delta = term ? (r - qsa) : (r + gamma * qmax - qsa);
            const float sc1 = lr * delta;
            const float sc2 = lr * (-gamma * td_est);
            const float sc3 = gp.lr_td * (delta - td_est);
            float sb1[A], sb2[A], sb3[A];
#pragma unroll
            for (int b = 0; b < A; ++b) {
                sb1[b] = (a == b) ? sc1 : 0.0f;
                sb2[b] = (!term && na_star == b) ? sc2 : 0.0f;
                sb3[b] = (a == b) ? sc3 : 0.0f;
            }
            w.axpy(sb1, phi_s);
            w.axpy(sb2, phi_n);
            v.axpy(sb3, phi_s);
            // ---- behaviour_policy.sample with the UPDATED fa_q (two columns may have moved: recompute)
            w.q(phi_n, q_n);
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            int na = policy_sample<A>(pol, q_n, x);
            facc_abs += fabsf(delta); facc_r += r;
            if (term) { n_ep += 1; sum_len += ep; ep = 0; }
            if (trunc) {
                n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
                Dom::reset(ns);
                { float ph[F]; Bas::project(ns, ph); phi_n.set(ph); }
                w.q(phi_n, q_n);
                const U4 xr = draw(c.seed, gid, t, BLK_RESET);
                na = policy_sample<A>(pol, q_n, xr);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
#pragma unroll
            for (int b = 0; b < A; ++b) q_s[b] = q_n[b];
            a = na;
        };
        int k = 0;
        for (; k + 1 < n_steps; k += 2) {
            one_step(phi_a, phi_b, t0 + (uint64_t)k);
            one_step(phi_b, phi_a, t0 + (uint64_t)k + 1);
        }
        if (k < n_steps) one_step(phi_a, phi_b, t0 + (uint64_t)k);
        sum_abs = (double)facc_abs; sum_r = (double)facc_r;
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) {
                c.W[((int64_t)(b * F + f)) * N + i] = w.get(b, f);
                gp.V[((int64_t)(b * F + f)) * N + i] = v.get(b, f);
            }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// Handler<&Transition>::handle of GreedyGQ on caller-supplied transitions (W and V in memory)
template <int DOMAIN, int ORDER>
__global__ __launch_bounds__(kBlock) void k_handle_gq(Common c, GqParams gp, const float* __restrict__ from,
                                                      const int32_t* __restrict__ act, const float* __restrict__ rew,
                                                      const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                      int64_t Mn, float* __restrict__ td_out) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    const int64_t N = c.n_envs;
    float s[D], ns[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { s[d] = from[(int64_t)d * Mn + i]; ns[d] = to[(int64_t)d * Mn + i]; }
    const int a = clamp_action<Dom::A>(act[i]);
    const float r = rew[i];
    const bool term = termf[i] != 0;
    float phi_s[F], phi_n[F], q_s[A], q_n[A], e_s[A];
    Bas::project(s, phi_s);
    Bas::project(ns, phi_n);
    q_from_mem<A, F>(c.W, N, i, phi_s, q_s);
    q_from_mem<A, F>(c.W, N, i, phi_n, q_n);
    q_from_mem<A, F>(gp.V, N, i, phi_s, e_s);
    const float qsa = select_a<A>(q_s, a), td_est = select_a<A>(e_s, a);
    float qmax;
    const int na_star = find_max<A>(q_n, qmax);
    const float delta = term ? (r - qsa) : (r + c.alg.gamma * qmax - qsa);
    const float sc1 = c.alg.lr * delta;
    const float sc2 = c.alg.lr * (-c.alg.gamma * td_est);
    const float sc3 = gp.lr_td * (delta - td_est);
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int64_t ja = ((int64_t)(a * F + f)) * N + i;
        c.W[ja] = fmaf(sc1, phi_s[f], c.W[ja]);
        gp.V[ja] = fmaf(sc3, phi_s[f], gp.V[ja]);
    }
    if (!term) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int64_t jn = ((int64_t)(na_star * F + f)) * N + i;
            c.W[jn] = fmaf(sc2, phi_n[f], c.W[jn]);
        }
    }
    if (td_out) td_out[i] = delta;
}

// ---- GreedyGQ over ANY model (tile coding with per-learner tables, the generic Fourier orders): both approximators in memory.
// The reference's agent is generic over the approximator (greedy_gq.rs:49-71: `Q` and `W` are type parameters); fa_td lives in the ctx's
// auxiliary matrix, addressed like fa_q (a Common whose weight pointer is the auxiliary matrix).  Operation by operation the register
// family's step -- and the oracle's orc_handle_gq, which is basis-generic: fa_q's column a moves by lr*td_error*phi(s) FIRST, then
// column na by lr*(-gamma*td_est)*phi(s') (non-terminal), fa_td's column a by lr_td*(td_error - td_est)*phi(s).
template <class M>
__device__ __forceinline__ float gq_handle_mem(const Common& c, const Common& cv, const GqParams& gp, const BasisGeom& g, int64_t i, const typename M::Feat& fs,
                                               int a, float r, const typename M::Feat& fn, bool term) {
    constexpr int A = M::A;
    float q_s[A], e_s[A], q_n[A];
    M::q_all(c, i, g, fs, q_s);
    M::q_all(cv, i, g, fs, e_s);
    M::q_all(c, i, g, fn, q_n);                               // (a terminal transition does not read it: the select below drops it)
    const float qsa = select_a<A>(q_s, a), td_est = select_a<A>(e_s, a);
    float qmax;
    const int na_star = find_max<A>(q_n, qmax);
    const float delta = term ? (r - qsa) : (r + c.alg.gamma * qmax - qsa);
    M::update(c, i, g, fs, a, c.alg.lr * delta);
    if (!term) M::update(c, i, g, fn, na_star, c.alg.lr * (-c.alg.gamma * td_est));
    M::update(cv, i, g, fs, a, gp.lr_td * (delta - td_est));
    return delta;
}
template <class M>
__global__ __launch_bounds__(kBlock) void k_train_gq_mem(Common c, GqParams gp, BasisGeom g, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
    constexpr int D = M::D, A = M::A;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
        Common cv = c; cv.W = gp.V;                            // fa_td: the same layout, the auxiliary matrix
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        float s[D]; load_state<M>(c.state, N, i, s);
        int a = c.action[i];
        uint32_t ep = c.ep_step[i];
        typename M::Feat fs, fn;
        M::features(s, g, fs);
        float facc_abs = 0.0f, facc_r = 0.0f;
        for (int k = 0; k < n_steps; ++k) {
            const uint64_t t = t0 + (uint64_t)k;
            float ns[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            const bool term = M::Dom::step(ns, a, r);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            M::features(ns, g, fn);
            const float delta = gq_handle_mem<M>(c, cv, gp, g, i, fs, a, r, fn, term);
            if (term || trunc) {
                n_ep += 1; n_trunc += trunc ? 1 : 0; sum_len += ep; ep = 0;
                M::Dom::reset(ns);
                M::features(ns, g, fn);
            }
            float q_n[A];
            M::q_all(c, i, g, fn, q_n);                        // behaviour policy: the UPDATED fa_q, at s' or at the restart state
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            a = policy_sample<A>(c.pol, q_n, x);
            facc_abs += fabsf(delta); facc_r += r;
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
            fs = fn;
        }
        sum_abs = (double)facc_abs; sum_r = (double)facc_r;
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}
template <class M>
__global__ __launch_bounds__(kBlock) void k_handle_gq_mem(Common c, GqParams gp, BasisGeom g, const float* __restrict__ from, const int32_t* __restrict__ act,
                                                          const float* __restrict__ rew, const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                          int64_t Mn, float* __restrict__ td_out) {
    constexpr int D = M::D, A = M::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    Common cv = c; cv.W = gp.V;
    float s[D], ns[D];
    load_state<M>(from, Mn, i, s);
    load_state<M>(to, Mn, i, ns);
    typename M::Feat fs, fn;
    M::features(s, g, fs);
    M::features(ns, g, fn);
    const float delta = gq_handle_mem<M>(c, cv, gp, g, i, fs, clamp_action<A>(act[i]), rew[i], fn, termf[i] != 0);
    if (td_out) td_out[i] = delta;
}

}  // namespace rsrl
