// kernels_lambda.hpp -- eligibility-trace control on the register family:
//   SARSALambda::handle   rsrl/src/control/td/sarsa_lambda.rs:53-98
//   QLambda::handle       rsrl/src/control/td/q_lambda.rs:56-99  (Watkins: the trace is cut when the action taken was
//                                                                not argmax_first of Q(s,.), utils.rs:23-34)
//   trace update rules    rsrl/src/traces.rs:188-240   Accumulate z = gl*z + g | Saturate clip(gl*z + g, -1, 1) |
//                                                      Dutch z = gl*(1-alpha)*z + g       (g = phi(s) in column a)
//   weight update         Handler<ScaledGradientUpdate>: W += (alpha*residual)*Z, bypassing the optimiser
//                         (rsrl/src/fa/linear.rs:184-196; examples/sarsa_lambda.rs:30 uses SGD(1.0))
// The trace Z has the shape of W and the same HBM layout (f32[A][F][N], learner fastest).  In the fused driver loop W
// AND Z stay in registers for the whole launch (2 x 108 for MountainCar Fourier(5)).  A step-cap truncation does not
// reset the trace (the reference only resets on a terminal transition).
#pragma once

#include "models.hpp"

namespace rsrl {

enum : int { ALG_SARSA_LAMBDA = 3, ALG_Q_LAMBDA = 4 };
enum : int { TRACE_ACCUMULATE = 0, TRACE_SATURATE = 1, TRACE_DUTCH = 2 };

struct LambdaParams {
    float* Z;          // [A][F][N]
    float rate;        // gamma*lambda (Dutch: * (1 - alpha))
    float alpha;       // step size of the lambda agents
    int trace;         // TRACE_*
};

// z <- rule(rate_eff * z + g)
__device__ __forceinline__ float trace_merge(int rule, float rate_eff, float z, float g) {
    float v = fmaf(rate_eff, z, g);
    if (rule == TRACE_SATURATE) v = fmaxf(-1.0f, fminf(1.0f, v));
    return v;
}

template <int DOMAIN, int ORDER, int ALGO, int POLICY>
__global__ __launch_bounds__(kBlock) void k_train_lambda(Common c, LambdaParams lp, uint64_t t0, int n_steps,
                                                         DevStats* __restrict__ stats) {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    constexpr int D = Dom::D, A = Dom::A, F = Bas::F;
    static_assert(ALGO == ALG_SARSA_LAMBDA || ALGO == ALG_Q_LAMBDA, "lambda agents only");
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = c.n_envs;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
        PolicyParams pol = c.pol; pol.kind = POLICY;
        const bool esched = c.eps != nullptr;                          // wave-uniform: the per-learner epsilon schedule (examples/sarsa_lambda.rs:68)
        if (esched) learner_eps_load(c, i, pol);
        AlgoParams alg = c.alg; alg.kind = (ALGO == ALG_SARSA_LAMBDA) ? ALG_SARSA : ALG_QLEARNING;   // the TD target formula
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        float s[D];
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
        int a = c.action[i];
        uint32_t ep = c.ep_step[i];
        constexpr bool PK = (RSRL_PK != 0) && (F % 4 == 0);
        using Phi = PhiBuf<F, PK>;
        WBuf<A, F, PK> w, z;
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) {
                w.put(b, f, c.W[((int64_t)(b * F + f)) * N + i]);
                z.put(b, f, lp.Z[((int64_t)(b * F + f)) * N + i]);
            }
        Phi phi_a;
        float q_s[A];
        { float ph[F]; Bas::project(s, ph); phi_a.set(ph); }
        w.q(phi_a, q_s);
        float facc_abs = 0.0f, facc_r = 0.0f;
        bool cut = false;          // trace.reset() of a terminal transition, applied as a zero decay rate at the next update

        // ONE feature buffer: the trace update is the only consumer of phi(s) and needs nothing of this step but the action, so it
        // runs FIRST and phi(s') then overwrites the buffer (W and Z are 216 of the 256 architectural registers: a second
        // 36-value buffer was paid for in AGPR copies)
        auto one_step = [&](Phi& phi, uint64_t t) {
            const float qsa = select_a<A>(q_s, a);
            // ---- trace: (Q(lambda): cut unless the action was the greedy one) then z = rule(rate*z + grad)
            float rate_eff = cut ? 0.0f : lp.rate;
            if constexpr (ALGO == ALG_Q_LAMBDA) rate_eff = (a != argmax_first<A>(q_s)) ? 0.0f : rate_eff;
            {
                float ind[A];
#pragma unroll
                for (int b = 0; b < A; ++b) ind[b] = (a == b) ? 1.0f : 0.0f;
                z.decay_add(rate_eff, ind, phi);
                if (lp.trace == TRACE_SATURATE) z.clip(-1.0f, 1.0f);
            }
            float ns[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            const bool term = Dom::step(ns, a, r);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            if (term) Dom::reset(ns);
            float q_n[A];
            Phi& phi_n = phi;
            { float ph[F]; Bas::project(ns, ph); phi_n.set(ph); }
            w.q(phi_n, q_n);
            // ---- residual with the PRE-update weights
            U4 xin = U4{0, 0, 0, 0};
            if constexpr (ALGO == ALG_SARSA_LAMBDA) xin = draw(c.seed, gid, t, BLK_INNER);
            float e;
            float delta;          // SARSALambda owns its policy (sarsa_lambda.rs:37-44): the inner draw comes from it
            if (ALGO == ALG_SARSA_LAMBDA && !c.apol_same) delta = td_error<A>(alg, c.apol, qsa, q_n, r, term, xin, e);
            else delta = td_error<A>(alg, pol, qsa, q_n, r, term, xin, e);
            // ---- W += (alpha * residual) * Z ; a terminal transition then resets the trace
            w.axpy_buf(lp.alpha * delta, z);
            cut = term;
            // ---- policy.sample with the UPDATED weights (every column moved: recompute)
            w.q(phi_n, q_n);
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            if (esched) learner_eps_step(c, term | trunc, pol);        // agent.policy.epsilon *= decay after the episode's last handle (:68)
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
            one_step(phi_a, t0 + (uint64_t)k);
            one_step(phi_a, t0 + (uint64_t)k + 1);
        }
        if (k < n_steps) one_step(phi_a, t0 + (uint64_t)k);
        sum_abs = (double)facc_abs; sum_r = (double)facc_r;
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
        if (esched) c.eps[i] = pol.eps;
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int f = 0; f < F; ++f) {
                c.W[((int64_t)(b * F + f)) * N + i] = w.get(b, f);
                lp.Z[((int64_t)(b * F + f)) * N + i] = cut ? 0.0f : z.get(b, f);
            }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// Handler<&Transition>::handle of the lambda agents on caller-supplied transitions (W and Z in memory)
template <int DOMAIN, int ORDER>
__global__ __launch_bounds__(kBlock) void k_handle_lambda(Common c, LambdaParams lp, const float* __restrict__ from,
                                                          const int32_t* __restrict__ act, const float* __restrict__ rew,
                                                          const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                          int64_t Mn, uint64_t t, float* __restrict__ td_out) {
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
    float phi_s[F], phi_n[F], q_s[A], q_n[A];
    Bas::project(s, phi_s);
    Bas::project(ns, phi_n);
    q_from_mem<A, F>(c.W, N, i, phi_s, q_s);
    q_from_mem<A, F>(c.W, N, i, phi_n, q_n);
    AlgoParams alg = c.alg;
    const bool sarsa = c.alg.kind == ALG_SARSA_LAMBDA;
    alg.kind = sarsa ? ALG_SARSA : ALG_QLEARNING;
    float rate_eff = lp.rate;
    if (!sarsa) rate_eff = (a != argmax_first<A>(q_s)) ? 0.0f : lp.rate;
    U4 xin = U4{0, 0, 0, 0};
    if (sarsa) xin = draw(c.seed, (uint32_t)(c.env_offset + i), t, BLK_INNER);
    float e;
    PolicyParams apol = c.apol;
    if (c.apol_same) learner_eps_load(c, i, apol);                     // the shared policy object: this learner's epsilon
    const float delta = td_error<A>(alg, apol, select_a<A>(q_s, a), q_n, r, term, xin, e);
    const float scale = lp.alpha * delta;
#pragma unroll
    for (int b = 0; b < A; ++b)
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int64_t j = ((int64_t)(b * F + f)) * N + i;
            const float zz = trace_merge(lp.trace, rate_eff, lp.Z[j], (a == b) ? phi_s[f] : 0.0f);
            c.W[j] = fmaf(scale, zz, c.W[j]);
            lp.Z[j] = term ? 0.0f : zz;
        }
    if (td_out) td_out[i] = delta;
}

}  // namespace rsrl
