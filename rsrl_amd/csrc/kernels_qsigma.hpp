// kernels_qsigma.hpp -- QSigma: the n-step Q(sigma) agent (De Asis et al. 2017, arXiv:1703.01327).
//   QSigma::handle        rsrl/src/control/td/q_sigma.rs:138-201
//   QSigma::update_backup rsrl/src/control/td/q_sigma.rs:107-128
//   Backup::propagate     rsrl/src/control/td/q_sigma.rs:46-63
//
// DEVIATION FROM THE REFERENCE (documented, one line): Backup::propagate loops `for k in 0..n_steps` and reads
// entries[k + 1] (q_sigma.rs:52-53) while update_backup calls it as soon as len == n_steps (:113-114), so its last
// iteration indexes entries[n_steps] out of bounds and the reference PANICS at the first full backup.  The only thing
// that iteration needs entries[k + 1] for is the update of z (:56) -- a value that is never used again.  The repair keeps
// every in-bounds operation of the loop (g += z*residual_k and the isr factor of ALL n entries) and drops that dead z update
// at k = n_steps - 1.  n_steps = 1 is then one-step Q(sigma) (sigma = 1: SARSA's target with an importance ratio, sigma = 0:
// the greedy tree backup).
//
// Per learner: a ring buffer of n_steps backup entries {s, a, q, residual, pi, mu} in memory (SoA, learner fastest:
// buf[field][slot][N]; sigma is the agent's constant), head / length counters, weights W f32[A][F][N] in memory (three
// projections per step -- s, s', the anchor's s -- and a column update at the ANCHOR's state: nothing to carry in registers).
// The agent samples its own a' from its own policy (q_sigma.rs:157: thread_rng) = draw block BLK_INNER, like SARSA.
//   pi = 1/|argmaxima| if a' is a maximum else 0 (:162-166: the TARGET policy is greedy, exp_nqs is the maximum)
//   mu = policy.evaluate((s', a')) (:167): the probability for Greedy / EpsilonGreedy / Random, and -- faithfully -- the raw
//        Q(s', a') for Softmax, whose Function<(S, A)> returns the action value (softmax.rs:84-92)
#pragma once

#include "models.hpp"

namespace rsrl {

enum : int { ALG_QSIGMA = 9 };

struct QsParams {
    float* buf;        // [D + 5][n_steps][N]: s[0..D), a (int bits), q, residual, pi, mu
    uint32_t* head;    // [N] ring position of the oldest entry
    uint32_t* len;     // [N] entries held
    int n_steps;
    float sigma, alpha;
};

// QSigma::handle on one transition of learner i (W row stride = c.w_stride, learner offset wi).  Returns the one-step residual
// pushed into the backup (the quantity the reference computes per transition); the weight update, when the backup is full,
// happens at the ANCHOR.  Model M = FourierModel<DOMAIN, ORDER>.
template <class M>
__device__ __forceinline__ float qsigma_handle(const Common& c, const QsParams& qp, const BasisGeom& g, int64_t i, int64_t N, const float (&s)[M::D], int a,
                                               float r, const float (&ns)[M::D], bool term, const U4& xin) {
    constexpr int D = M::D, A = M::A;
    typename M::Feat fs;
    M::features(s, g, fs);
    const float qa = M::q_index(c, i, g, fs, a);                                 // :140
    float residual, pi, mu;
    if (term) {
        residual = r - qa; pi = 0.0f; mu = 1.0f;                                 // :142-153
    } else {
        typename M::Feat fn;
        M::features(ns, g, fn);
        float nqs[A];
        M::q_all(c, i, g, fn, nqs);
        const int na = policy_sample<A>(c.apol, nqs, xin);                       // :157 the agent's own draw
        const float nqsna = select_a<A>(nqs, na);
        float exp_nqs;
        const uint32_t mask = argmaxima_mask_max<A>(nqs, exp_nqs);               // :161
        pi = ((mask >> na) & 1u) ? 1.0f / (float)__popc(mask) : 0.0f;            // :163-167
        mu = policy_eval_sa<A>(c.apol, nqs, na);                                 // :168
        residual = r + c.alg.gamma * (qp.sigma * nqsna + (1.0f - qp.sigma) * exp_nqs) - qa;    // :170-171
    }
    // ---- update_backup (:107-128): push, then one update of the anchor once n_steps entries are held
    const int n = qp.n_steps;
    uint32_t head = qp.head[i], len = qp.len[i];
    const int64_t fs_stride = (int64_t)n * N;                                    // between fields
    auto at = [&](int field, uint32_t slot) -> float& { return qp.buf[(int64_t)field * fs_stride + (int64_t)slot * N + i]; };
    {
        const uint32_t slot = (head + len) % (uint32_t)n;
#pragma unroll
        for (int d = 0; d < D; ++d) at(d, slot) = s[d];
        at(D, slot) = __int_as_float(a);
        at(D + 1, slot) = qa; at(D + 2, slot) = residual; at(D + 3, slot) = pi; at(D + 4, slot) = mu;
        len += 1;
    }
    if ((int)len >= n) {
        // Backup::propagate (:46-63) with the dead out-of-bounds z update of the last iteration dropped (header)
        float gret = at(D + 1, head), z = 1.0f, isr = 1.0f;
        for (int k = 0; k < n; ++k) {
            const uint32_t s1 = (head + (uint32_t)k) % (uint32_t)n;
            gret += z * at(D + 2, s1);
            if (k + 1 < n) {
                const uint32_t s2 = (head + (uint32_t)k + 1u) % (uint32_t)n;
                z *= c.alg.gamma * ((1.0f - qp.sigma) * at(D + 3, s2) + qp.sigma);
            }
            isr *= 1.0f - qp.sigma + qp.sigma * at(D + 3, s1) / at(D + 4, s1);
        }
        float as_[D];
#pragma unroll
        for (int d = 0; d < D; ++d) as_[d] = at(d, head);
        const int aa = clamp_action<A>(__float_as_int(at(D, head)));
        head = (head + 1u) % (uint32_t)n; len -= 1;                              // pop (:116)
        typename M::Feat fa;
        M::features(as_, g, fa);
        const float qsa = M::q_index(c, i, g, fa, aa);                           // :117 with the CURRENT weights
        const float err = qp.alpha * isr * (gret - qsa);                         // :122
        M::update(c, i, g, fa, aa, c.alg.lr * err);                              // Handler<StateActionUpdate>: W[:,a] += lr*error*phi
    }
    if (term) len = 0;                                                           // backup.clear() (:154)
    qp.head[i] = head; qp.len[i] = len;
    return residual;
}

// the driver loop (examples/q_learning.rs:34-55 with a QSigma agent): weights in memory, n_steps batch-steps per launch
template <class M>
__global__ __launch_bounds__(kBlock) void k_train_qsigma(Common c, QsParams qp, BasisGeom g, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
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
            const U4 xin = draw(c.seed, gid, t, BLK_INNER);
            const float res = qsigma_handle<M>(c, qp, g, i, N, s, a, r, ns, term, xin);
            if (term || trunc) {
                n_ep += 1; n_trunc += trunc ? 1 : 0; sum_len += ep; ep = 0;
                M::Dom::reset(ns);
            }
            // policy.sample with the UPDATED weights at s' (or at s0 after the episode ended)
            typename M::Feat fn; float q_n[A];
            M::features(ns, g, fn);
            M::q_all(c, i, g, fn, q_n);
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            a = policy_sample<A>(c.pol, q_n, x);
            facc_abs += fabsf(res); facc_r += r;
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
        }
        sum_abs = (double)facc_abs; sum_r = (double)facc_r;
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// Handler<&Transition>::handle on caller-supplied transitions (item m = learner m)
template <class M>
__global__ __launch_bounds__(kBlock) void k_handle_qsigma(Common c, QsParams qp, BasisGeom g, const float* __restrict__ from, const int32_t* __restrict__ act,
                                                          const float* __restrict__ rew, const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                          int64_t Mn, uint64_t t, float* __restrict__ td_out) {
    constexpr int D = M::D, A = M::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    float s[D], ns[D];
    load_state<M>(from, Mn, i, s);
    load_state<M>(to, Mn, i, ns);
    const U4 xin = draw(c.seed, (uint32_t)(c.env_offset + i), t, BLK_INNER);
    const float res = qsigma_handle<M>(c, qp, g, i, c.n_envs, s, clamp_action<A>(act[i]), rew[i], ns, termf[i] != 0, xin);
    if (td_out) td_out[i] = res;
}

}  // namespace rsrl
