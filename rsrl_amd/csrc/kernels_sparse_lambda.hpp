// kernels_sparse_lambda.hpp -- eligibility traces over ONE SHARED tile-coded table (round 5): SARSALambda / QLambda with
// weight_mode = RSRL_W_SHARED, basis = RSRL_TILE_CODING.
//
// The reference's Trace<B, R> is generic over its buffer (rsrl/src/traces.rs:5-12) and ships a sparse one (rsrl/src/params/sparse.rs:13-97): a
// learner's trace over a tile-coded table is non-zero only where the learner has recently been.  Here every learner owns such a SPARSE trace --
// a list of (key = tile index * A + action, value) -- next to the one table all learners share; a batch-step is the synchronous mini-batch rule
// of the other shared-weight modes:
//     every learner i, against the same W_t:   Q(s,.), Q(s',.), residual_i (SARSA: the agent's own draw; Q: max), its trace update
//                                              z_i <- rule(rate * z_i + g),  g = 1 at the T active entries of column a
//                                                     (Q(lambda): z_i cleared first when a was not argmax_first of Q(s,.), q_lambda.rs:62-66)
//     W_{t+1} = W_t + sum_i (alpha * residual_i) * z_i          terminal transition: z_i <- 0 afterwards (sarsa_lambda.rs:91-93)
//     then every learner samples its next action from W_{t+1}; finished episodes restart (a step cap does not reset the trace)
// The sum over learners runs in 64-bit fixed point (lsb = 2^(floor(log2 alpha) - 28), FxScale): exact, order-independent, reproducible.
//
// The list.  kSparseCap = 512 entries per learner (the reference's buffer grows without bound; 512 entries are 64 steps of 8 tilings that never
// revisit a tile -- by then an accumulating trace has decayed to rate^64).  One WAVE per learner: slot s lives in lane s & 63, register s >> 6.
// Per step, in this order (restated one for one by the oracle, orc_run_train_sparse_lambda):
//   1. every entry: v <- rule(fma(rate, v, hit ? 1 : 0)), hit = its key is one of the step's T new keys (the dense rule on the non-zero entries);
//   2. the new keys that were not in the list, in tiling order: appended at slot len (value rule(fma(rate, 0, 1)) = 1) -- or, when the list is
//      full, written over the entry with the smallest |v| (ties: the lowest slot), which is how the cap takes effect;
//   3. the learner's terms (alpha * residual) * v go to the fixed-point table at their keys; a terminal transition then empties the list.
// Entries are never removed otherwise (a value that decays to a denormal stays, as a dense trace keeps it).
#pragma once

#include "models.hpp"
#include "kernels_lambda.hpp"

namespace rsrl {

constexpr int kSparseCap = 512, kSparseRegs = kSparseCap / 64;

struct SparseTrace {
    uint32_t* keys;    // [N][kSparseCap]
    float* vals;       // [N][kSparseCap]
    uint32_t* len;     // [N]
};
// what phase A hands to phase C: the successor state (before any restart), reward / flags
struct SparseMail { float* ns; uint8_t* flags; };      // ns [D][N]; flags: bit 0 terminal, bit 1 truncated

// wave-wide (min |v|, slot) as one 64-bit key: |v| bits above the slot, so the smallest value wins and ties go to the lowest slot
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

// phase A: one wave per learner
template <int DOMAIN, int T>
__global__ __launch_bounds__(kBlock) void k_sparse_lambda_step(Common c, BasisGeom g, LambdaParams lp, SparseTrace st, SparseMail mail,
                                                               long long* __restrict__ fx, uint64_t t, DevStats* __restrict__ stats) {
    using M = TileModel<DOMAIN, T>;
    using Dom = Domain<DOMAIN>;
    constexpr int D = M::D, A = M::A;
    const int lane = threadIdx.x & 63;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
        const bool sarsa = c.alg.kind == ALG_SARSA_LAMBDA;
        AlgoParams alg = c.alg; alg.kind = sarsa ? ALG_SARSA : ALG_QLEARNING;      // the TD target formula
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        float s[D], ns[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { s[d] = c.state[(int64_t)d * N + i]; ns[d] = s[d]; }
        const int a = __builtin_amdgcn_readfirstlane(c.action[i]);
        float r;
        const bool term = Dom::step(ns, a, r);
        const uint32_t ep = c.ep_step[i] + 1;
        const bool trunc = !term && c.max_episode_steps > 0 && ep >= c.max_episode_steps;
        typename M::Feat fs, fn;
        M::features(s, g, fs);
        M::features(ns, g, fn);
        float q_s[A], q_n[A];
        M::q_all(c, 0, g, fs, q_s);
        M::q_all(c, 0, g, fn, q_n);
        // ---- the trace
        uint32_t* __restrict__ K = st.keys + i * (int64_t)kSparseCap;
        float* __restrict__ V = st.vals + i * (int64_t)kSparseCap;
        int len = (int)st.len[i];
        if (!sarsa && a != argmax_first<A>(q_s)) len = 0;                          // Watkins's cut
        uint32_t key[kSparseRegs]; float val[kSparseRegs];
#pragma unroll
        for (int e = 0; e < kSparseRegs; ++e) {
            const int slot = e * 64 + lane;
            key[e] = slot < len ? K[slot] : 0xffffffffu;
            val[e] = slot < len ? V[slot] : 0.0f;
        }
        uint32_t nk[T]; bool found[T];
#pragma unroll
        for (int tt = 0; tt < T; ++tt) { nk[tt] = (uint32_t)fs.idx[tt] * A + (uint32_t)a; found[tt] = false; }
#pragma unroll
        for (int e = 0; e < kSparseRegs; ++e) {
            bool hit = false;
#pragma unroll
            for (int tt = 0; tt < T; ++tt) {
                const bool m = key[e] == nk[tt];
                hit = hit || m;
                found[tt] = found[tt] || (__ballot(m) != 0ull);
            }
            val[e] = trace_merge(lp.trace, lp.rate, val[e], hit ? 1.0f : 0.0f);
        }
        const float fresh = trace_merge(lp.trace, lp.rate, 0.0f, 1.0f);
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
            if (found[tt]) continue;                                               // (wave-uniform)
            int slot;
            if (len < kSparseCap) { slot = len; len += 1; }
            else {
                unsigned long long best = ~0ull;
#pragma unroll
                for (int e = 0; e < kSparseRegs; ++e) {
                    const unsigned long long cand = ((unsigned long long)(__builtin_bit_cast(uint32_t, val[e]) & 0x7fffffffu) << 32) | (uint32_t)(e * 64 + lane);
                    best = cand < best ? cand : best;
                }
                slot = (int)(uint32_t)wave_min_u64(best);
            }
#pragma unroll
            for (int e = 0; e < kSparseRegs; ++e)
                if (slot == e * 64 + lane) { key[e] = nk[tt]; val[e] = fresh; }
        }
        // ---- residual against W_t, the learner's terms into the fixed-point table
        U4 xin = U4{0, 0, 0, 0};
        if (sarsa) xin = draw(c.seed, gid, t, BLK_INNER);                          // the agent's own draw (sarsa_lambda.rs:78)
        float e_;
        const float delta = td_error<A>(alg, c.apol, select_a<A>(q_s, a), q_n, r, term, xin, e_);
        const float scale = lp.alpha * delta;
        const float inv_lsb = FxScale(lp.alpha).inv_lsb;
#pragma unroll
        for (int e = 0; e < kSparseRegs; ++e)
            if (e * 64 + lane < len) fx_add(&fx[key[e]], fx_quantise(scale * val[e], inv_lsb));
        if (term) len = 0;                                                         // trace.reset()
#pragma unroll
        for (int e = 0; e < kSparseRegs; ++e) {
            const int slot = e * 64 + lane;
            if (slot < len) { K[slot] = key[e]; V[slot] = val[e]; }
        }
        if (lane == 0) {
            st.len[i] = (uint32_t)len;
#pragma unroll
            for (int d = 0; d < D; ++d) mail.ns[(int64_t)d * N + i] = ns[d];
            mail.flags[i] = (uint8_t)((term ? 1 : 0) | (trunc ? 2 : 0));
            c.ep_step[i] = ep;
            sum_abs = (double)fabsf(delta); sum_r = (double)r;
        }
    }
    if (stats) block_stats_accumulate(stats, 0, 0, 0, sum_abs, sum_r);
}

// phase C: one thread per learner -- the behaviour policy's sample from W_{t+1}, restarts
template <int DOMAIN, int T>
__global__ __launch_bounds__(kBlock) void k_sparse_lambda_sample(Common c, BasisGeom g, SparseMail mail, uint64_t t, DevStats* __restrict__ stats) {
    using M = TileModel<DOMAIN, T>;
    using Dom = Domain<DOMAIN>;
    constexpr int D = M::D, A = M::A;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    if (i < N) {
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        float ns[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ns[d] = mail.ns[(int64_t)d * N + i];
        const uint8_t fl = mail.flags[i];
        uint32_t ep = c.ep_step[i];
        if (fl) {                                                                  // the episode ended: restart (step cap: the trace lives on)
            n_ep = 1; n_trunc = (fl & 2) ? 1 : 0; sum_len = ep; ep = 0;
            Dom::reset(ns);
        }
        typename M::Feat fn;
        M::features(ns, g, fn);
        float q[A];
        M::q_all(c, 0, g, fn, q);
        const U4 x = draw(c.seed, gid, t, BLK_STEP);                               // (BLK_RESET is the same draw: the step's one behaviour sample)
        const int na = policy_sample<A>(c.pol, q, x);
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = ns[d];
        c.action[i] = na;
        c.ep_step[i] = ep;
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, 0.0, 0.0);
}

// Parameterised-style view of one learner's trace: the dense (F, A) matrix it stands for (zeros + the list's entries)
__global__ void k_sparse_trace_get(SparseTrace st, int64_t i, float* __restrict__ out /* zero-filled [F][A] */) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < (int)st.len[i]) out[st.keys[i * (int64_t)kSparseCap + s]] = st.vals[i * (int64_t)kSparseCap + s];
}

}  // namespace rsrl
