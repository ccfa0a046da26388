// kernels_td.hpp -- prediction on the register family: the state-value analogue of the control path.
//   TD::handle         rsrl/src/prediction/td/td.rs:31-59
//   TDLambda::handle   rsrl/src/prediction/td/td_lambda.rs:41-78
//   ScalarLFA          rsrl/src/fa/linear.rs:201-251   V(s) = <phi(s), w>, grad = phi(s), StateUpdate -> w += lr*error*phi(s)
// One weight column per learner, f32[F][N] (learner fastest), register-resident in the fused loop like the control kernels.
// The driver loop is the reference's with a Random behaviour policy (the only policy that needs no Q):
//     t = env.transition(a);  agent.handle(&t);  a = Random.sample(rng)        (+ auto-reset, step cap)
//   TD       : td = r + gamma*V(s') - V(s)   (terminal: r - V(s));   w += lr * td * phi(s)
//   TDLambda : trace.update(phi(s)) first (traces.rs:188-240), then  w += td * trace  -- ScaledGradientUpdate{alpha: td_error}
//              (td_lambda.rs:59-62): the step IS the TD error, no learning rate; a terminal transition resets the trace.
#pragma once

#include "models.hpp"
#include "kernels_lambda.hpp"

namespace rsrl {

enum : int { ALG_TD = 7, ALG_TD_LAMBDA = 8 };

struct TdParams {
    float* Z;          // trace [F][N] (TDLambda)
    float rate;        // gamma*lambda (Dutch: * (1 - alpha))
    int trace;         // TRACE_*
};

template <int DOMAIN, int ORDER, bool LAMBDA>
__global__ __launch_bounds__(kBlock) void k_train_td(Common c, TdParams tp, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = c.n_envs;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
        PolicyParams pol = c.pol; pol.kind = POL_RANDOM;
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
        WBuf<1, F, PK> w, z;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            w.put(0, f, c.W[(int64_t)f * N + i]);
            z.put(0, f, LAMBDA ? tp.Z[(int64_t)f * N + i] : 0.0f);
        }
        Phi phi_a, phi_b;
        float v_s[1];
        { float ph[F]; Bas::project(s, ph); phi_a.set(ph); }
        w.q(phi_a, v_s);
        float facc_abs = 0.0f, facc_r = 0.0f;
        bool cut = false;
        const float q0[A] = {};                        // Random ignores the action values

        auto one_step = [&](const Phi& phi_s, Phi& phi_n, uint64_t t) {
            float ns[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            const bool term = Dom::step(ns, a, r);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            if (term) Dom::reset(ns);
            float v_n[1];
            { float ph[F]; Bas::project(ns, ph); phi_n.set(ph); }
            w.q(phi_n, v_n);
            if constexpr (LAMBDA) {
                const float one[1] = {1.0f};
                z.decay_add(cut ? 0.0f : tp.rate, one, phi_s);
                if (tp.trace == TRACE_SATURATE) z.clip(-1.0f, 1.0f);
            }
            const float td = term ? (r - v_s[0]) : (r + gamma * v_n[0] - v_s[0]);
            if constexpr (LAMBDA) {
                w.axpy_buf(td, z);
                cut = term;
            } else {
                const float sb[1] = {lr * td};
                w.axpy(sb, phi_s);
            }
            w.q(phi_n, v_n);                              // V(s') with the UPDATED weights: the next step's prediction
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            int na = policy_sample<A>(pol, q0, x);
            facc_abs += fabsf(td); facc_r += r;
            if (term) { n_ep += 1; sum_len += ep; ep = 0; }
            if (trunc) {
                n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
                Dom::reset(ns);
                { float ph[F]; Bas::project(ns, ph); phi_n.set(ph); }
                w.q(phi_n, v_n);
                const U4 xr = draw(c.seed, gid, t, BLK_RESET);
                na = policy_sample<A>(pol, q0, xr);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
            v_s[0] = v_n[0];
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
        for (int f = 0; f < F; ++f) {
            c.W[(int64_t)f * N + i] = w.get(0, f);
            if constexpr (LAMBDA) tp.Z[(int64_t)f * N + i] = cut ? 0.0f : z.get(0, f);
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// Handler<&Transition>::handle of TD / TDLambda on caller-supplied transitions (w and z in memory); lambda: runtime flag
template <int DOMAIN, int ORDER>
__global__ __launch_bounds__(kBlock) void k_handle_td(Common c, TdParams tp, int lambda, const float* __restrict__ from,
                                                      const float* __restrict__ rew, const float* __restrict__ to,
                                                      const uint8_t* __restrict__ termf, int64_t Mn, float* __restrict__ td_out) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, F = Bas::F;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    const int64_t N = c.n_envs;
    float s[D], ns[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { s[d] = from[(int64_t)d * Mn + i]; ns[d] = to[(int64_t)d * Mn + i]; }
    const float r = rew[i];
    const bool term = termf[i] != 0;
    float phi_s[F], phi_n[F], v_s[1], v_n[1];
    Bas::project(s, phi_s);
    Bas::project(ns, phi_n);
    q_from_mem<1, F>(c.W, N, i, phi_s, v_s);
    q_from_mem<1, F>(c.W, N, i, phi_n, v_n);
    const float td = term ? (r - v_s[0]) : (r + c.alg.gamma * v_n[0] - v_s[0]);
    const float scale = c.alg.lr * td;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int64_t j = (int64_t)f * N + i;
        if (lambda) {
            const float zz = trace_merge(tp.trace, tp.rate, tp.Z[j], phi_s[f]);
            c.W[j] = fmaf(td, zz, c.W[j]);
            tp.Z[j] = term ? 0.0f : zz;
        } else {
            c.W[j] = fmaf(scale, phi_s[f], c.W[j]);
        }
    }
    if (td_out) td_out[i] = td;
}

// Function<(S,)>::evaluate of the ScalarLFA: V(s_i) with learner i's weights
template <int DOMAIN, int ORDER>
__global__ __launch_bounds__(kBlock) void k_v_evaluate(Common c, const float* __restrict__ states, int64_t Mn, float* __restrict__ out) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, F = Bas::F;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    float s[D], phi[F], v[1];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = states[(int64_t)d * Mn + i];
    Bas::project(s, phi);
    q_from_mem<1, F>(c.W, c.n_envs, i, phi, v);
    out[i] = v[0];
}

// TD / TDLambda on a model WITHOUT a register-family kernel (the generic Fourier orders): one thread per learner, w and z in memory
// ([F][N], learner fastest), phi recomputed per feature (M::phi_at).  from == nullptr: the driver loop; otherwise Handler::handle on
// one caller-supplied transition per learner.  Same arithmetic as orc_handle_td (oracle/rsrl_oracle_impl.h): V(s) by M::q_index on
// column 0, the trace merged before the weights move, a terminal transition zeroing the trace.
template <class M>
__global__ __launch_bounds__(kBlock) void k_td_mem(Common c, TdParams tp, BasisGeom g, int lambda, uint64_t t0, int n_steps, DevStats* __restrict__ stats,
                                                   const float* __restrict__ from, const float* __restrict__ rew, const float* __restrict__ to,
                                                   const uint8_t* __restrict__ termf, int64_t Mn, float* __restrict__ td_out) {
    constexpr int D = M::D, A = M::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = c.n_envs;
    const bool driver = from == nullptr;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < (driver ? N : Mn)) {
        PolicyParams pol = c.pol; pol.kind = POL_RANDOM;
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        const float q0[A] = {};
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
            typename M::Feat fs, fn;
            M::features(s, g, fs);
            M::features(ns, g, fn);
            const float v_s = M::q_index(c, i, g, fs, 0);
            const float v_n = M::q_index(c, i, g, fn, 0);
            const float td = term ? (r - v_s) : (r + c.alg.gamma * v_n - v_s);
            if (lambda) {
                for (int f = 0; f < g.F; ++f) {
                    const int64_t j = M::widx(c, i, g, 0, f);
                    const float zz = trace_merge(tp.trace, tp.rate, tp.Z[j], M::phi_at(g, fs, f));
                    c.W[j] = fmaf(td, zz, c.W[j]);
                    tp.Z[j] = term ? 0.0f : zz;
                }
            } else {
                M::update(c, i, g, fs, 0, c.alg.lr * td);
            }
            if (!driver) { if (td_out) td_out[i] = td; break; }
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            int na = policy_sample<A>(pol, q0, x);
            sum_abs += (double)fabsf(td); sum_r += (double)r;
            if (term) { n_ep += 1; sum_len += ep; ep = 0; }
            if (trunc) {
                n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
                M::Dom::reset(ns);
                const U4 xr = draw(c.seed, gid, t, BLK_RESET);
                na = policy_sample<A>(pol, q0, xr);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
            a = na;
        }
        if (driver) {
#pragma unroll
            for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
            c.action[i] = a;
            c.ep_step[i] = ep;
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// Function<(S,)>::evaluate of the ScalarLFA on such a model
template <class M>
__global__ __launch_bounds__(kBlock) void k_v_mem(Common c, BasisGeom g, const float* __restrict__ states, int64_t Mn, float* __restrict__ out) {
    constexpr int D = M::D;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    float s[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = states[(int64_t)d * Mn + i];
    typename M::Feat ft;
    M::features(s, g, ft);
    out[i] = M::q_index(c, i, g, ft, 0);
}

// per-episode Domain::default() + the first Random.sample (the control path's k_reset evaluates Q, which does not exist here)
template <int DOMAIN>
__global__ __launch_bounds__(kBlock) void k_reset_td(Common c, uint64_t t) {
    using Dom = Domain<DOMAIN>;
    constexpr int D = Dom::D, A = Dom::A;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float s[D];
    Dom::reset(s);
#pragma unroll
    for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
    PolicyParams pol = c.pol; pol.kind = POL_RANDOM;
    const float q0[A] = {};
    const U4 x = draw(c.seed, (uint32_t)(c.env_offset + i), t, BLK_INIT);
    c.action[i] = policy_sample<A>(pol, q0, x);
    c.ep_step[i] = 0;
}

}  // namespace rsrl
