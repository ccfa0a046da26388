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


// ---------------------------------------------------------------------------------------------------------------------------------------
// The driver loop again, FOUR threads per learner (round 6; VERDICT r5 item 5: k_train_lambda_mem ran at 0.03 of 8 TB/s on its algorithmic bytes -- one thread
// per learner walks 5 x A x F dependent memory accesses per step with a D-digit decode + D - 1 complex products per feature, and 16 384 learners are a
// quarter of a wave per SIMD).
//
// What the arithmetic allows without moving a bit: Q(s,.) is FOUR interleaved partial sums (features f = p mod 4 in ascending order, combined
// (a0 + a1) + (a2 + a3): FourierGenericModel::q_all), so four threads can own one partial each; the sweep over the (F, A) entries is element-wise.  A block is
// 64 learners x 4 partials: wave p holds partial p of the block's 64 learners, lane = learner, so every access stays one contiguous line of the learner-fastest
// layout.  Per batch-step:
//   tables   wave d builds dimension d's harmonic table of s' (the angle-addition chain of FourierGenericModel::features) into LDS, shared by the four waves;
//   pass 1   ONE read of the partial's rows of W for Q(s,.) AND Q(s',.) (the old loop read W twice for them); partials exchanged through LDS;
//   pass 2   the sweep z = rule(rate z + g), W += (alpha residual) z over the partial's rows, and IN THE SAME PASS Q(s'',.) of the state the behaviour policy
//            samples at (s', or the restart state) from the weights it has just written -- the dot product a fresh evaluation would form, in its order
//            (the old loop read W a third time for it);
//   features feature f's value is the real part of a product over the dimensions in order (phi_at); f walks with stride 4, the mixed-radix digits are carried
//            along instead of decoded by D divisions, and the product over dimensions 0 .. D-2 is kept until a higher digit moves: ~1 complex product per
//            feature instead of D - 1 -- the same operations on the same operands, cached.
// W is read twice and written once per step, Z once each: 5 A F -> the algorithmic 4 A F + one extra read of W.
// ---------------------------------------------------------------------------------------------------------------------------------------
template <int D>
struct FeatWalk {                 // the features f = p, p + 4, p + 8, ... of one partial, for up to three table sets at once
    int c[D];                     // digits of k = f + 1, c[0] most significant (phi_at)
    int n1, f, F;
    bool pre_stale;
    __device__ __forceinline__ void start(int p, int order, int F_) {
        n1 = order + 1; F = F_; f = p;
        int k = p + 1;
#pragma unroll
        for (int d = D - 1; d >= 0; --d) { c[d] = k % n1; k /= n1; }
        pre_stale = true;
    }
    __device__ __forceinline__ void advance() {                       // f += 4
        f += 4;
        c[D - 1] += 4;
        if (c[D - 1] >= n1) {
            pre_stale = true;
#pragma unroll
            for (int d = D - 1; d > 0; --d) {
                while (c[d] >= n1) { c[d] -= n1; c[d - 1] += 1; }
            }
        }
    }
};
// tab[set][d][n][0 = cos, 1 = sin][lane]
template <int D>
__device__ __forceinline__ void walk_prefix(const float* __restrict__ tab, int set, int lane, const FeatWalk<D>& w, float& re, float& im) {
    auto T = [&](int d, int n, int cs) { return tab[((((size_t)set * D + d) * 8 + n) * 2 + cs) * 64 + lane]; };
    re = T(0, w.c[0], 0); im = T(0, w.c[0], 1);
#pragma unroll
    for (int d = 1; d < D - 1; ++d) {
        const float cr = T(d, w.c[d], 0), sr = T(d, w.c[d], 1);
        const float nre = fmaf(-im, sr, re * cr), nim = fmaf(re, sr, im * cr);
        re = nre; im = nim;
    }
}
template <int D>
__device__ __forceinline__ float walk_phi(const float* __restrict__ tab, int set, int lane, const FeatWalk<D>& w, float pre_re, float pre_im) {
    if (w.f == w.F - 1) return 1.0f;                                  // the constant feature (with_bias)
    if constexpr (D == 1) return pre_re;
    const float cr = tab[((((size_t)set * D + (D - 1)) * 8 + w.c[D - 1]) * 2 + 0) * 64 + lane];
    const float sr = tab[((((size_t)set * D + (D - 1)) * 8 + w.c[D - 1]) * 2 + 1) * 64 + lane];
    return fmaf(-pre_im, sr, pre_re * cr);
}

template <class M>
__global__ __launch_bounds__(256) void k_train_lambda_mem4(Common c, LambdaParams lp, BasisGeom g, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
    using Dom = typename M::Dom;
    constexpr int D = M::D, A = M::A;
    static_assert(D >= 2 && D <= 4, "one wave per dimension builds the tables");
    __shared__ float tab[3 * D * 8 * 2 * 64];        // table sets 0 / 1: s and s' in ping-pong; set 2: the restart state's
    __shared__ float xq[2][4][2 * A][64];            // the partial sums of the two exchanges
    const int64_t N = c.n_envs;
    const int lane = (int)(threadIdx.x & 63), p = (int)(threadIdx.x >> 6);
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const int64_t il = i < N ? i : N - 1;             // lanes beyond N repeat the last learner and store nothing (they take part in the barriers)
    const bool live = i < N;
    const int order = g.tiles_per_dim, F = g.F;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    float facc_abs = 0.0f, facc_r = 0.0f;
    const uint32_t gid = (uint32_t)(c.env_offset + il);
    const uint32_t cap = c.max_episode_steps;
    const bool sarsa = c.alg.kind == ALG_SARSA_LAMBDA;
    AlgoParams alg = c.alg; alg.kind = sarsa ? ALG_SARSA : ALG_QLEARNING;
    float s[D]; load_state<M>(c.state, N, il, s);
    int a = c.action[il];
    uint32_t ep = c.ep_step[il];
    float* __restrict__ const Wp = c.W;
    float* __restrict__ const Zp = lp.Z;

    // dimension `dim`'s harmonic table of state x into table set `set` (FourierGenericModel::features, one dimension)
    auto build = [&](int set, const float (&x)[D], int dim) {
        static_for<0, D>([&](auto Dd) {
            constexpr int d = Dd;
            if (d != dim) return;
            constexpr float lo = (float)Dom::lo_d(d), hi = (float)Dom::hi_d(d);
            constexpr float inv = 1.0f / (hi - lo);
            const float sc = (x[d] - lo) * inv;
            float ct[8], st[8];
            ct[0] = 1.0f; st[0] = 0.0f;
            sincospi01(sc, st[1], ct[1]);
#pragma unroll
            for (int n = 2; n < 8; ++n) {
                ct[n] = fmaf(-st[n - 1], st[1], ct[n - 1] * ct[1]);
                st[n] = fmaf(ct[n - 1], st[1], st[n - 1] * ct[1]);
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                tab[((((size_t)set * D + d) * 8 + n) * 2 + 0) * 64 + lane] = ct[n];
                tab[((((size_t)set * D + d) * 8 + n) * 2 + 1) * 64 + lane] = st[n];
            }
        });
    };
    int cur = 0;
    if (p < D) {
        build(0, s, p);
        float s0[D]; Dom::reset(s0);
        build(2, s0, p);
    }
    __syncthreads();

    for (int k = 0; k < n_steps; ++k) {
        const uint64_t t = t0 + (uint64_t)k;
        float ns[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ns[d] = s[d];
        float r;
        const bool term = Dom::step(ns, a, r);
        ep += 1;
        const bool trunc = !term && cap > 0 && ep >= cap;
        const int nxt = cur ^ 1;
        if (p < D) build(nxt, ns, p);
        __syncthreads();
        // ---- pass 1: this partial's share of Q(s,.) and Q(s',.) from ONE read of its rows
        float acc_s[A], acc_n[A];
#pragma unroll
        for (int b = 0; b < A; ++b) { acc_s[b] = 0.0f; acc_n[b] = 0.0f; }
        {
            FeatWalk<D> w; w.start(p, order, F);
            float ps_re = 0.0f, ps_im = 0.0f, pn_re = 0.0f, pn_im = 0.0f;
#pragma unroll 16
            for (; w.f < F; w.advance()) {
                if (w.pre_stale && w.f != F - 1) { walk_prefix<D>(tab, cur, lane, w, ps_re, ps_im); walk_prefix<D>(tab, nxt, lane, w, pn_re, pn_im); w.pre_stale = false; }
                const float phs = walk_phi<D>(tab, cur, lane, w, ps_re, ps_im), phn = walk_phi<D>(tab, nxt, lane, w, pn_re, pn_im);
#pragma unroll
                for (int b = 0; b < A; ++b) {
                    const float wv = Wp[((int64_t)b * F + w.f) * N + il];
                    acc_s[b] = fmaf(phs, wv, acc_s[b]);
                    acc_n[b] = fmaf(phn, wv, acc_n[b]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < A; ++b) { xq[0][p][b][lane] = acc_s[b]; xq[0][p][A + b][lane] = acc_n[b]; }
        __syncthreads();
        float q_s[A], q_n[A];
#pragma unroll
        for (int b = 0; b < A; ++b) {
            q_s[b] = (xq[0][0][b][lane] + xq[0][1][b][lane]) + (xq[0][2][b][lane] + xq[0][3][b][lane]);
            q_n[b] = (xq[0][0][A + b][lane] + xq[0][1][A + b][lane]) + (xq[0][2][A + b][lane] + xq[0][3][A + b][lane]);
        }
        // ---- the residual (every wave of the block the same: lambda_handle_mem)
        float rate_eff = lp.rate;
        if (!sarsa) rate_eff = (a != argmax_first<A>(q_s)) ? 0.0f : lp.rate;
        U4 xin = U4{0, 0, 0, 0};
        if (sarsa) xin = draw(c.seed, gid, t, BLK_INNER);
        float e;
        const float delta = td_error<A>(alg, c.apol, select_a<A>(q_s, a), q_n, r, term, xin, e);
        const float scale = lp.alpha * delta;
        // ---- the state the behaviour policy samples at: s', or the restart state (its tables replace s' 's for the learners that restart)
        const bool restart = term || trunc;
        if (restart) {
            n_ep += 1; n_trunc += trunc ? 1 : 0; sum_len += ep; ep = 0;
            Dom::reset(ns);
            if (p < D) {
#pragma unroll
                for (int n = 0; n < 8; ++n)
#pragma unroll
                    for (int cs = 0; cs < 2; ++cs)
                        tab[((((size_t)nxt * D + p) * 8 + n) * 2 + cs) * 64 + lane] = tab[((((size_t)2 * D + p) * 8 + n) * 2 + cs) * 64 + lane];
            }
        }
        __syncthreads();
        // ---- pass 2: the sweep over this partial's rows + its share of Q(s'',.) under the weights it writes
        float acc_q[A];
#pragma unroll
        for (int b = 0; b < A; ++b) acc_q[b] = 0.0f;
        {
            FeatWalk<D> w; w.start(p, order, F);
            float ps_re = 0.0f, ps_im = 0.0f, pn_re = 0.0f, pn_im = 0.0f;
            constexpr int G = 8;                                        // features per group: every load of a group is issued before its first store
                                                                        // (pass 1 unrolled x 16, G = 8: 72.6 -> 69.7 us at 16 384 learners; x 16 / G = 16: 69.3)
            while (w.f < F) {
                float zv[G][A], wv[G][A], phs[G], phn[G]; int fi[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    fi[u] = w.f;
                    if (w.f < F) {
                        if (w.pre_stale && w.f != F - 1) { walk_prefix<D>(tab, cur, lane, w, ps_re, ps_im); walk_prefix<D>(tab, nxt, lane, w, pn_re, pn_im); w.pre_stale = false; }
                        phs[u] = walk_phi<D>(tab, cur, lane, w, ps_re, ps_im); phn[u] = walk_phi<D>(tab, nxt, lane, w, pn_re, pn_im);
#pragma unroll
                        for (int b = 0; b < A; ++b) { const int64_t j = ((int64_t)b * F + w.f) * N + il; zv[u][b] = Zp[j]; wv[u][b] = Wp[j]; }
                        w.advance();
                    }
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    if (fi[u] < F) {
#pragma unroll
                        for (int b = 0; b < A; ++b) {
                            const int64_t j = ((int64_t)b * F + fi[u]) * N + il;
                            const float zz = trace_merge(lp.trace, rate_eff, zv[u][b], (a == b) ? phs[u] : 0.0f);
                            const float wn = fmaf(scale, zz, wv[u][b]);
                            if (live) { Wp[j] = wn; Zp[j] = term ? 0.0f : zz; }
                            acc_q[b] = fmaf(phn[u], wn, acc_q[b]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < A; ++b) xq[1][p][b][lane] = acc_q[b];
        __syncthreads();
        float q_p[A];
#pragma unroll
        for (int b = 0; b < A; ++b) q_p[b] = (xq[1][0][b][lane] + xq[1][1][b][lane]) + (xq[1][2][b][lane] + xq[1][3][b][lane]);
        const U4 x = draw(c.seed, gid, t, BLK_STEP);
        a = policy_sample<A>(c.pol, q_p, x);
        facc_abs += fabsf(delta); facc_r += r;
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = ns[d];
        cur = nxt;
    }
    if (live && p == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
    }
    const bool counts = live && p == 0;                                // a learner's statistics once, not four times
    if (stats) block_stats_accumulate(stats, counts ? n_ep : 0, counts ? n_trunc : 0, counts ? sum_len : 0, counts ? (double)facc_abs : 0.0, counts ? (double)facc_r : 0.0);
}

}  // namespace rsrl
