// kernels_lambda_mem.hpp -- eligibility-trace control (SARSALambda / QLambda) on the Fourier orders WITHOUT a register-family kernel
// (MountainCar 6-7, CartPole / Acrobot 2-6; round 5: the last agent x per-learner-family combination that was refused).
//
// The reference's agents are generic over the approximator (sarsa_lambda.rs:37-52, q_lambda.rs:37-54), its traces over the buffer
// (traces.rs:6-12).  Here W and the trace Z (same shape, same layout: rows of N learners) live in memory and one thread owns one learner,
// like the other *_mem kernels (GreedyGQ, QSigma, TD on these orders): per step Q(s,.) and Q(s',.) from W_t, the TD error (SARSA(lambda):
// the agent's own draw; Q(lambda): max, and the trace cut when the action taken was not argmax_first of Q(s,.), q_lambda.rs:62-66), then ONE
// sweep over the (F, A) entries:  z = rule(rate * z + g),  W += (alpha * residual) * z,  Z = z (terminal: 0) -- operation by operation
// orc_handle_lambda (oracle/rsrl_oracle_impl.h), so every weight and trace entry is bit-identical to the CPU run.
#pragma once

#include "kernels_lambda.hpp"

namespace rsrl {

template <class M>
__device__ __forceinline__ float lambda_handle_mem(const Common& c, const LambdaParams& lp, const BasisGeom& g, int64_t i, uint32_t gid, uint64_t t,
                                                   const typename M::Feat& fs, int a, float r, const typename M::Feat& fn, bool term) {
    constexpr int A = M::A;
    float q_s[A], q_n[A];
    M::q_all(c, i, g, fs, q_s);
    M::q_all(c, i, g, fn, q_n);
    const bool sarsa = c.alg.kind == ALG_SARSA_LAMBDA;
    AlgoParams alg = c.alg; alg.kind = sarsa ? ALG_SARSA : ALG_QLEARNING;      // the TD target formula
    float rate_eff = lp.rate;
    if (!sarsa) rate_eff = (a != argmax_first<A>(q_s)) ? 0.0f : lp.rate;
    U4 xin = U4{0, 0, 0, 0};
    if (sarsa) xin = draw(c.seed, gid, t, BLK_INNER);                          // the agent's own draw (sarsa_lambda.rs:78)
    float e;
    const float delta = td_error<A>(alg, c.apol, select_a<A>(q_s, a), q_n, r, term, xin, e);
    const float scale = lp.alpha * delta;
    // the sweep, EIGHT features at a time: every load of the group is issued before the first store (W and Z are distinct allocations, which the
    // compiler cannot know: written one entry at a time each iteration waits out a full memory round trip -- 622 -> 557 us per batch-step at 16 384 CartPole
    // learners of order 3: the three Q evaluations with their per-feature index arithmetic are the rest; a generic fallback, not a tuned kernel.
    // The values and their order per entry are the same)
    float* __restrict__ const Wp = c.W;
    float* __restrict__ const Zp = lp.Z;
    constexpr int G = 8;
    for (int f0 = 0; f0 < g.F; f0 += G) {
        float zv[G][A], wv[G][A], ph[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int f = f0 + u < g.F ? f0 + u : g.F - 1;          // (the tail repeats the last feature's loads; nothing of it is stored)
            ph[u] = M::phi_at(g, fs, f);
#pragma unroll
            for (int b = 0; b < A; ++b) { const int64_t j = M::widx(c, i, g, b, f); zv[u][b] = Zp[j]; wv[u][b] = Wp[j]; }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (f0 + u < g.F) {
#pragma unroll
                for (int b = 0; b < A; ++b) {
                    const int64_t j = M::widx(c, i, g, b, f0 + u);
                    const float zz = trace_merge(lp.trace, rate_eff, zv[u][b], (a == b) ? ph[u] : 0.0f);
                    Wp[j] = fmaf(scale, zz, wv[u][b]);
                    Zp[j] = term ? 0.0f : zz;
                }
            }
        }
    }
    return delta;
}

template <class M>
__global__ __launch_bounds__(kBlock) void k_train_lambda_mem(Common c, LambdaParams lp, BasisGeom g, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
    constexpr int D = M::D, A = M::A;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
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
            const float delta = lambda_handle_mem<M>(c, lp, g, i, gid, t, fs, a, r, fn, term);
            if (term || trunc) {                                   // (a step cap restarts the episode; only a terminal transition resets the trace)
                n_ep += 1; n_trunc += trunc ? 1 : 0; sum_len += ep; ep = 0;
                M::Dom::reset(ns);
                M::features(ns, g, fn);
            }
            float q_n[A];
            M::q_all(c, i, g, fn, q_n);                            // behaviour policy: the UPDATED weights, at s' or at the restart state
            const U4 x = draw(c.seed, gid, t, BLK_STEP);            // (BLK_RESET is the same draw: the step's one behaviour sample)
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

// Handler<&Transition>::handle on ONE caller-supplied transition per learner (teacher forcing)
template <class M>
__global__ __launch_bounds__(kBlock) void k_handle_lambda_mem(Common c, LambdaParams lp, BasisGeom g, const float* __restrict__ from, const int32_t* __restrict__ act,
                                                              const float* __restrict__ rew, const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                              int64_t Mn, uint64_t t, float* __restrict__ td_out) {
    constexpr int D = M::D, A = M::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    float s[D], ns[D];
    load_state<M>(from, Mn, i, s);
    load_state<M>(to, Mn, i, ns);
    typename M::Feat fs, fn;
    M::features(s, g, fs);
    M::features(ns, g, fn);
    const float delta = lambda_handle_mem<M>(c, lp, g, i, (uint32_t)(c.env_offset + i), t, fs, clamp_action<A>(act[i]), rew[i], fn, termf[i] != 0);
    if (td_out) td_out[i] = delta;
}

}  // namespace rsrl
