// kernels_lambda_tile.hpp -- eligibility-trace control (SARSALambda / QLambda) on TILE CODING, per-learner tables.
//
// The reference's traces are generic over the gradient buffer (rsrl/src/traces.rs:6-12; update rules :188-240): with a
// tile-coding basis the gradient of Q(s, a) is 1.0 at the T active entries of column a, so
//     z <- rule(rate * z + g),   g = 1 at the T active entries of column a, 0 elsewhere        (every entry decays)
//     W <- W + (alpha * residual) * z                                                          (fa/linear.rs:184-196)
//     terminal transition: z <- 0                                                              (sarsa_lambda.rs:91-93, q_lambda.rs:93-95)
// on a dense trace table of W's shape (f32[N][F][A], the tile layout) per learner.  That is 2 x F*A values read and written per
// learner-step (1 MiB at 8 tilings x 8^4 x 2), so this kernel is a memory sweep: ONE BLOCK PER LEARNER, the learner's scalar
// work (transition, the 2 x T gathers of Q(s,.) and Q(s',.), the TD error, the policy) computed redundantly by every thread
// (wave-uniform: no divergence, no broadcast), then one fused, coalesced 16-byte sweep over (Z, W) with the non-active rule
// z = rule(rate * z + 0), and the T active entries -- whose old values were set aside before the sweep -- redone with g = 1.
// Element by element these are the oracle's operations (orc_handle_lambda), so W and Z are bit-identical to the CPU run.
// Order of a batch-step: k_train_mem's (models.hpp) = the reference's (examples/sarsa_lambda.rs:48-75).
#pragma once

#include "kernels_lambda.hpp"

namespace rsrl {

// The table sweep uses non-temporal loads AND stores: a learner's 1 MiB comes back a whole batch-step later (2 048 CartPole learners, env-steps/s:
// plain 5.8-5.9e6, loads nt 5.99e6, stores nt 5.87e6, both 6.15-6.23e6; about one process in four runs 15-20 % slower with byte-identical code --
// where its 1 GiB of tables landed).

// from == nullptr: the driver loop, n_steps batch-steps of learner blockIdx.x.  Otherwise Handler::handle on ONE caller-supplied
// transition per learner (teacher forcing), Mn learners.
template <int DOMAIN, int T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_lambda_tile(Common c, BasisGeom g, LambdaParams lp, uint64_t t0, int n_steps, DevStats* __restrict__ stats,
                                                       const float* __restrict__ from, const int32_t* __restrict__ act, const float* __restrict__ rew,
                                                       const float* __restrict__ to, const uint8_t* __restrict__ termf, int64_t Mn,
                                                       float* __restrict__ td_out) {
    using M = TileModel<DOMAIN, T>;
    constexpr int D = M::D, A = M::A;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t i = blockIdx.x;
    const int64_t N = c.n_envs;
    const int tid = (int)threadIdx.x;
    const int FA = g.F * A;
    float* __restrict__ Wl = c.W + i * (int64_t)FA;
    float* __restrict__ Zl = lp.Z + i * (int64_t)FA;
    Common cl = c; cl.W = Wl;                                   // TileModel::q_all(cl, 0, ..) addresses this learner's table
    const bool sarsa = c.alg.kind == ALG_SARSA_LAMBDA;
    AlgoParams alg = c.alg; alg.kind = sarsa ? ALG_SARSA : ALG_QLEARNING;      // the TD target formula
    const uint32_t gid = (uint32_t)(c.env_offset + i);
    const uint32_t cap = c.max_episode_steps;
    const bool driver = from == nullptr;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;

    float s[D];
    int a; uint32_t ep = 0;
    if (driver) {
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
        a = c.action[i]; ep = c.ep_step[i];
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = from[(int64_t)d * Mn + i];
        a = clamp_action<A>(act[i]);
    }
    typename M::Feat fs, fn;
    M::features(s, g, fs);
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
        float q_s[A], q_n[A];
        M::q_all(cl, 0, g, fs, q_s);
        M::q_all(cl, 0, g, fn, q_n);
        // ---- trace decay rate: Q(lambda) cuts the trace unless the action taken was argmax_first of Q(s,.)   q_lambda.rs:62-66
        float rate_eff = lp.rate;
        if (!sarsa) rate_eff = (a != argmax_first<A>(q_s)) ? 0.0f : lp.rate;
        U4 xin = U4{0, 0, 0, 0};
        if (sarsa) xin = draw(c.seed, gid, t, BLK_INNER);                      // the agent's own draw (sarsa_lambda.rs:78)
        float e;
        const float delta = td_error<A>(alg, c.apol, select_a<A>(q_s, a), q_n, r, term, xin, e);
        const float scale = lp.alpha * delta;
        // ---- the T active entries: old values set aside (the sweep treats every entry as non-active)
        float zo = 0.0f, wo = 0.0f; int ea = 0;
        if (tid < T) {
            int idx = fs.idx[0];
#pragma unroll
            for (int tt = 1; tt < T; ++tt) idx = (tid == tt) ? fs.idx[tt] : idx;
            ea = idx * A + a;
            zo = Zl[ea]; wo = Wl[ea];
        }
        __syncthreads();                                        // every gather of W above precedes every store below
        for (int j = tid * 4; j < FA; j += BLOCK * 4) {
            const f4 z4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Zl + j));
            f4 w4 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Wl + j));
            f4 zz;
            zz.x = trace_merge(lp.trace, rate_eff, z4.x, 0.0f); zz.y = trace_merge(lp.trace, rate_eff, z4.y, 0.0f);
            zz.z = trace_merge(lp.trace, rate_eff, z4.z, 0.0f); zz.w = trace_merge(lp.trace, rate_eff, z4.w, 0.0f);
            w4.x = fmaf(scale, zz.x, w4.x); w4.y = fmaf(scale, zz.y, w4.y); w4.z = fmaf(scale, zz.z, w4.z); w4.w = fmaf(scale, zz.w, w4.w);
            __builtin_nontemporal_store(w4, reinterpret_cast<f4*>(Wl + j));
            __builtin_nontemporal_store(term ? f4{0.0f, 0.0f, 0.0f, 0.0f} : zz, reinterpret_cast<f4*>(Zl + j));
        }
        __syncthreads();
        if (tid < T) {
            const float zz = trace_merge(lp.trace, rate_eff, zo, 1.0f);
            Wl[ea] = fmaf(scale, zz, wo);
            Zl[ea] = term ? 0.0f : zz;
        }
        __syncthreads();                                        // the table is complete before anybody gathers from it again
        if (!driver) { if (tid == 0 && td_out) td_out[i] = delta; break; }
        // ---- policy.sample with the UPDATED weights
        M::q_all(cl, 0, g, fn, q_n);
        const U4 x = draw(c.seed, gid, t, BLK_STEP);
        int na = policy_sample<A>(c.pol, q_n, x);
        sum_abs += (double)fabsf(delta); sum_r += (double)r;
        if (term) { n_ep += 1; sum_len += ep; ep = 0; }
        if (trunc) {                                             // step cap: new episode; the trace is NOT reset (only a terminal does)
            n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
            M::Dom::reset(ns);
            M::features(ns, g, fn);
            M::q_all(cl, 0, g, fn, q_n);
            const U4 xr = draw(c.seed, gid, t, BLK_RESET);
            na = policy_sample<A>(c.pol, q_n, xr);
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
    if (stats) {                                                 // one slot per block = per learner; only thread 0 contributes
        const bool me = tid == 0;
        block_stats_accumulate(stats, me ? n_ep : 0ull, me ? n_trunc : 0ull, me ? sum_len : 0ull, me ? sum_abs : 0.0, me ? sum_r : 0.0);
    }
}

}  // namespace rsrl
