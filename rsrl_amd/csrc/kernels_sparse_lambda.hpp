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
// The list.  kSparseCap = 512 entries per learner (the reference's buffer grows without bound), held as T SUB-LISTS, one per tiling, of
// kSparseCap / T entries each (64 entries per tiling at T = 8: 64 steps that never revisit a tile of that tiling -- by then an accumulating trace
// has decayed to rate^64).  A key belongs to exactly one tiling (key / (cells * A)), a step brings at most one new key per tiling, and every
// operation on the trace -- decay, hit, append, evict, the learner's terms, reset -- acts on the tilings independently.  So the unit of work is
// (learner, tiling), and the trace update runs INSIDE the scatter kernel that already owns a tiling's slice of the delta table in LDS:
//
//   k_shared_ca   (models.hpp, one thread per learner; the one-step shared-table agents' step kernel)   phase C of the previous batch-step
//                 (policy.sample from the updated table, restarts) + phase A of this one: transition, Q(s,.), Q(s',.), the TD target's residual;
//                 hands over alpha * residual, the T new slice-relative keys, and flags (bit 0 terminal, bit 1 truncated, bit 2 Watkins's cut)
//   k_sparse_trace_scatter   block (chunk of learners, tiling t), one WAVE per (learner, tiling) at a time: slot s of the sub-list lives in
//                 lane s & 63, register s >> 6.  In this order (restated one for one by the oracle, orc_run_train_sparse_lambda):
//                   0. Q(lambda) and a was not argmax_first of Q(s,.): the sub-list is emptied first (q_lambda.rs:62-66);
//                   1. every entry: v <- rule(fma(rate, v, hit ? 1 : 0)), hit = its key is the step's new key of this tiling;
//                   2. the new key, if it was not in the sub-list: appended (value rule(fma(rate, 0, 1))) -- or, when the sub-list is full, written
//                      over its entry with the smallest |v| (ties: the lowest slot), which is how the cap takes effect;
//                   3. the learner's terms (alpha * residual) * v into the tiling's LDS slice (64-bit fixed point: exact, any order);
//                      a terminal transition then empties the sub-list (sarsa_lambda.rs:91-93).
//                 One sweep of the slice at the end: one device atomic per touched entry into one of n_rep copies of the table.
//   k_apply_rep   W += fl(sum of the copies * lsb), copies cleared (multi-rank: the float delta goes through the exchange first).
// Round 5's form -- one wave per LEARNER with the transition replicated over its 64 lanes, <= 512 device atomics per learner-step, four
// launches -- ran at 246 us per batch-step at 16 384 CartPole learners; this one: profiles/r06_new_kernels.md.
// Entries are never removed otherwise (a value that decays to a denormal stays, as a dense trace keeps it).
#pragma once

#include "models.hpp"
#include "kernels_lambda.hpp"

namespace rsrl {

constexpr int kSparseCap = 512;

struct SparseTrace {
    uint32_t* keys;    // [N][kSparseCap]: sub-list t in slots [t * cap_t, (t + 1) * cap_t); key = tile index * A + action
    float* vals;       // [N][kSparseCap]
    uint32_t* len;     // [N][T]
};

// wave-wide (min |v|, slot) as one 64-bit key: |v| bits above the slot, so the smallest value wins and ties go to the lowest slot
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

// S = entries of one tiling's slice (cells * A); lds != 0: the slice is privatised in dynamic LDS (S * 8 bytes), else the terms go straight to
// copy 0 of the table with device atomics (a slice too large for LDS: the same integers, the same sum).
template <int T>
__global__ __launch_bounds__(1024) void k_sparse_trace_scatter(const uint16_t* __restrict__ new_keys, const float* __restrict__ terms,
                                                               const uint8_t* __restrict__ flags, SparseTrace st, LambdaParams lp, int64_t N, int64_t key_stride,
                                                               int S, int per_block, long long* __restrict__ dW64, int n_rep, int64_t rep_stride,
                                                               float inv_lsb, int lds) {
    constexpr int CAP = kSparseCap / T, REGS = (CAP + 63) / 64, U = 4;      // sub-lists in flight per wave (8: 2-3x SLOWER, 74.8 / 310 us at 16 384 / 65 536 learners -- measured, round 6)
    extern __shared__ long long sparse_slice[];
    const int t = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * per_block;
    const int64_t i1 = i0 + per_block < N ? i0 + per_block : N;
    const uint16_t* __restrict__ kt = new_keys + (int64_t)t * key_stride;      // (N: the learners stepped, 0 .. N-1; key_stride: the ctx's learner count)
    const uint32_t base = (uint32_t)t * (uint32_t)S;                         // full key = base + slice-relative key
    long long* __restrict__ dst = dW64 + (int64_t)(lds ? blockIdx.x % (unsigned)n_rep : 0u) * rep_stride + (int64_t)t * S;
    if (lds) {
        for (int j = threadIdx.x; j < S; j += blockDim.x) sparse_slice[j] = 0;
        __syncthreads();
    }
    const float fresh = trace_merge(lp.trace, lp.rate, 0.0f, 1.0f);
    // the sub-lists' lengths run ONE batch ahead of their entries: from the second batch on a batch's loads touch only the live slots (a list is rarely full: the rest
    // of its 256-byte row is dead weight, and the kernel runs at the box's copy bandwidth) without a dependent round trip in front of them.  65 536 CartPole learners:
    // 122 -> 102 us per batch-step; at 16 384 (latency-bound: one block per CU) nothing to gain
    int len_next[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + wave + (int64_t)u * n_waves;
        len_next[u] = i < i1 ? (int)st.len[i * T + t] : 0;
    }
    for (int64_t ib = i0 + wave; ib < i1; ib += (int64_t)U * n_waves) {
        const bool first = ib == i0 + wave;
        // U learners' sub-lists in flight together
        uint32_t key[U][REGS]; float val[U][REGS]; int len[U]; uint32_t nk[U]; float sc[U]; uint8_t fl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = ib + (int64_t)u * n_waves;
            const bool ok = i < i1;
            const int64_t ii = ok ? i : i0;
            len[u] = __builtin_amdgcn_readfirstlane(len_next[u]);
            nk[u] = base + (uint32_t)kt[ii]; sc[u] = terms[ii]; fl[u] = flags[ii];
#pragma unroll
            for (int e = 0; e < REGS; ++e) {
                const int slot = e * 64 + lane;
                const bool in = slot < (first ? CAP : len[u]);                       // (len <= CAP; the first batch does not wait for its lengths)
                key[u][e] = in ? st.keys[ii * kSparseCap + t * CAP + slot] : 0xffffffffu;
                val[u][e] = in ? st.vals[ii * kSparseCap + t * CAP + slot] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = ib + (int64_t)(U + u) * n_waves;
            len_next[u] = i < i1 ? (int)st.len[i * T + t] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = ib + (int64_t)u * n_waves;
            if (i >= i1) break;                                                    // (wave-uniform)
            int ln = len[u];
            if (fl[u] & 4) ln = 0;                                                  // Watkins's cut
            bool found = false;
#pragma unroll
            for (int e = 0; e < REGS; ++e) {
                const bool live = e * 64 + lane < ln;
                const bool hit = live && key[u][e] == nk[u];
                found = found || (__ballot(hit) != 0ull);
                val[u][e] = trace_merge(lp.trace, lp.rate, live ? val[u][e] : 0.0f, hit ? 1.0f : 0.0f);
            }
            if (!found) {
                int slot;
                if (ln < CAP) { slot = ln; ln += 1; }
                else {
                    unsigned long long best = ~0ull;
#pragma unroll
                    for (int e = 0; e < REGS; ++e) {
                        if (e * 64 + lane >= CAP) continue;
                        const unsigned long long cand = ((unsigned long long)(__builtin_bit_cast(uint32_t, val[u][e]) & 0x7fffffffu) << 32) | (uint32_t)(e * 64 + lane);
                        best = cand < best ? cand : best;
                    }
                    slot = (int)(uint32_t)wave_min_u64(best);
                }
#pragma unroll
                for (int e = 0; e < REGS; ++e)
                    if (slot == e * 64 + lane) { key[u][e] = nk[u]; val[u][e] = fresh; }
            }
#pragma unroll
            for (int e = 0; e < REGS; ++e) {
                if (e * 64 + lane >= ln) continue;
                const unsigned long long q = fx_quantise(sc[u] * val[u][e], inv_lsb);
                if (q == 0) continue;
                if (lds) atomicAdd(reinterpret_cast<unsigned long long*>(&sparse_slice[key[u][e] - base]), q);
                else fx_add(&dst[key[u][e] - base], q);
            }
            if (fl[u] & 1) ln = 0;                                                  // trace.reset()
#pragma unroll
            for (int e = 0; e < REGS; ++e) {
                const int slot = e * 64 + lane;
                if (slot < ln) { st.keys[i * kSparseCap + t * CAP + slot] = key[u][e]; st.vals[i * kSparseCap + t * CAP + slot] = val[u][e]; }
            }
            if (lane == 0) st.len[i * T + t] = (uint32_t)ln;
        }
    }
    if (!lds) return;
    __syncthreads();
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const long long v = sparse_slice[j];
        if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&dst[j]), (unsigned long long)v);
    }
}

// Handler::handle of the sparse-trace agents (sarsa_lambda.rs:63-98, q_lambda.rs:56-99) on caller-supplied transitions: transition i is LEARNER i's (its
// trace is the one that moves; Mn <= the ctx's learners).  This kernel is k_shared_ca's phase A on the caller's (s, a, r, s', terminal): the residual
// against the shared table and the hand-over to k_sparse_trace_scatter; c.alg carries the TD target's formula and step size (as in the driver loop).
template <class M>
__global__ __launch_bounds__(kBlock) void k_sparse_handle(Common c, BasisGeom g, const float* __restrict__ from, const int32_t* __restrict__ act,
                                                          const float* __restrict__ rew, const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                          int64_t Mn, uint64_t t, int want_cut, uint8_t* __restrict__ flags, uint16_t* __restrict__ keys,
                                                          float* __restrict__ terms, float* __restrict__ td_out) {
    constexpr int D = M::D, A = M::A, T = M::kT;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    const uint32_t gid = (uint32_t)(c.env_offset + i);
    float s[D], ns[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { s[d] = from[(int64_t)d * Mn + i]; ns[d] = to[(int64_t)d * Mn + i]; }
    const int a = clamp_action<A>(act[i]);
    const float r = rew[i];
    const bool term = termf[i] != 0;
    typename M::Feat fs, fn;
    M::features(s, g, fs);
    M::features(ns, g, fn);
    float q_s[A], q_n[A];
    M::q_all_shared(c.W, g, fs, q_s);
    M::q_all_shared(c.W, g, fn, q_n);
    U4 xin = U4{0, 0, 0, 0};
    if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, gid, t, BLK_INNER);
    float e;
    const float delta = td_dispatch<A>(c.alg, c.apol, q_s, a, q_n, r, term, xin, e);
    const int cells = g.F / T;
#pragma unroll
    for (int tt = 0; tt < T; ++tt) keys[(int64_t)tt * c.n_envs + i] = (uint16_t)((fs.idx[tt] - tt * cells) * A + a);
    terms[i] = c.alg.lr * e;
    flags[i] = (uint8_t)((term ? 1 : 0) | ((want_cut && a != argmax_first<A>(q_s)) ? 4 : 0));
    if (td_out) td_out[i] = delta;
}

// Parameterised-style view of one learner's trace: the dense (F, A) matrix it stands for (zeros + the list's entries)
static __global__ void k_sparse_trace_get(SparseTrace st, int T, int64_t i, float* __restrict__ out /* zero-filled [F][A] */) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x, cap = kSparseCap / T;
    if (s < kSparseCap && s % cap < (int)st.len[i * T + s / cap]) out[st.keys[i * (int64_t)kSparseCap + s]] = st.vals[i * (int64_t)kSparseCap + s];
}

}  // namespace rsrl
