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
//   k_sparse_trace_scatter   block (chunk of learners, tiling t), one GROUP of 8 or 16 lanes per (learner, tiling) at a time (eight or four learners per wave): slot s of the
//                 sub-list lives in lane s % G of the group, register s / G.  In this order (restated one for one by the oracle, orc_run_train_sparse_lambda):
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
    uint16_t* keys;    // [N][kSparseCap]: sub-list t in slots [t * cap_t, (t + 1) * cap_t); a key is RELATIVE to its tiling's slice: (tile index - t * cells) * A + action
                       // (a slice holds at most 65 536 entries -- the step kernel's hand-over is 16 bit too --: 10 B of HBM traffic per live entry and step instead of 12)
    float* vals;       // [N][kSparseCap]
    uint32_t* len;     // [N][T]
};

// (min |v|, slot) over a G-lane group as one 64-bit key: |v| bits above the slot, so the smallest value wins and ties go to the lowest slot
template <int G>
__device__ __forceinline__ unsigned long long group_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
// the largest of a per-group value (equal over a group's G lanes) over the wave's 64 / G groups: wave-uniform
template <int G>
__device__ __forceinline__ int wave_max_of_groups(int v) {
    int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
    for (int g = 1; g < 64 / G; ++g) { const int x = __builtin_amdgcn_readlane(v, g * G); m = x > m ? x : m; }
    return m;
}
// fn(integral_constant<NR>) with NR the smallest of {1, 2, 4, ..., REGS} whose NR x G slots hold `slots` entries (slots is wave-uniform)
template <int REGS, int G, class F>
__device__ __forceinline__ void sparse_dispatch_regs(int slots, F&& fn) {
    if constexpr (REGS >= 2) { if (slots <= G) { fn(std::integral_constant<int, 1>{}); return; } }
    if constexpr (REGS >= 4) { if (slots <= 2 * G) { fn(std::integral_constant<int, 2>{}); return; } }
    if constexpr (REGS >= 8) { if (slots <= 4 * G) { fn(std::integral_constant<int, 4>{}); return; } }
    fn(std::integral_constant<int, REGS>{});
}

// S = entries of one tiling's slice (cells * A); lds != 0: the slice is privatised in dynamic LDS (S * 8 bytes), else the terms go straight to
// copy 0 of the table with device atomics (a slice too large for LDS: the same integers, the same sum).
//
// Mapping (round 6, second form): a sub-list belongs to a GROUP of G lanes -- slot s in lane s % G of the group, register s / G -- so a wave carries 64 / G
// learners' sub-lists of one tiling at a time and runs, wave-uniformly, only the registers the longest of them reaches.  A CartPole learner's sub-list holds ~11 live
// entries of 64: with one WAVE per sub-list (the first form) 53 of 64 lanes idled through every instruction and the kernel was issue-bound at a sixth of the lanes.
// G = 8 at 8 and 16 tilings (eight learners per wave; 65 536 learners: 50.9 -> 46.7 us per batch-step with short lists, 75.0 / 75.2 with full ones, against G = 16),
// 16 at 4 tilings (128 slots per sub-list would be sixteen registers).  Nothing about the values moves: every entry's arithmetic is its own, the eviction key orders
// (|v|, slot) as before, the sum is exact in any order.
template <int T>
__global__ __launch_bounds__(1024) void k_sparse_trace_scatter(const uint16_t* __restrict__ new_keys, const float* __restrict__ terms,
                                                               const uint8_t* __restrict__ flags, SparseTrace st, LambdaParams lp, int64_t N, int64_t key_stride,
                                                               int S, int per_block, long long* __restrict__ dW64, int n_rep, int64_t rep_stride,
                                                               float inv_lsb, int lds) {
    constexpr int CAP = kSparseCap / T, G = T >= 8 ? 8 : 16, REGS = CAP / G, LPW = 64 / G, U = REGS > 4 ? 2 : 4;   // U x LPW sub-lists in flight per wave
    static_assert(CAP % G == 0 && 64 % G == 0, "a sub-list is a whole number of G-slot registers, a wave a whole number of groups");
    extern __shared__ long long sparse_slice[];
    const int t = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int grp = lane / G, gl = lane % G;
    const int64_t i0 = (int64_t)blockIdx.x * per_block;
    const int64_t i1 = i0 + per_block < N ? i0 + per_block : N;
    const uint16_t* __restrict__ kt = new_keys + (int64_t)t * key_stride;      // (N: the learners stepped, 0 .. N-1; key_stride: the ctx's learner count)
    long long* __restrict__ dst = dW64 + (int64_t)(lds ? blockIdx.x % (unsigned)n_rep : 0u) * rep_stride + (int64_t)t * S;
    // everything below indexes the block's learners by a 32-bit number relative to i0 on wave-uniform base pointers (scalar base + 32-bit byte offset
    // addressing: a 64-bit address per array and sub-list in flight cost 124 VGPRs -- the whole register file of a CU for one 1 024-thread block, which a
    // co-tenant's waiting kernel could then keep from ever starting: tests/fuzz_ranks.py, G ranks on one device)
    const uint32_t nb = (uint32_t)(i1 - i0);                                 // learners of this block (<= per_block <= 2^20: byte offsets below stay under 2^32)
    const char* __restrict__ kbase = reinterpret_cast<const char*>(st.keys + i0 * kSparseCap + t * CAP);
    char* __restrict__ kbase_w = reinterpret_cast<char*>(st.keys + i0 * kSparseCap + t * CAP);
    char* __restrict__ vbase = reinterpret_cast<char*>(st.vals + i0 * kSparseCap + t * CAP);
    char* __restrict__ lbase = reinterpret_cast<char*>(st.len + i0 * T + t);
    const char* __restrict__ ktb = reinterpret_cast<const char*>(kt + i0);
    const char* __restrict__ tmb = reinterpret_cast<const char*>(terms + i0);
    const char* __restrict__ flb = reinterpret_cast<const char*>(flags + i0);
    const uint32_t ustride = (uint32_t)n_waves * LPW;                         // learners one u of the block covers
    // the sub-lists' lengths run ONE batch ahead of their entries (the first batch's are fetched under the clearing of the slice): a batch's loads touch only the
    // live slots -- a list is rarely full, the rest of its row is dead weight -- without a dependent round trip in front of them
    int len_next[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t r = (uint32_t)wave * LPW + grp + (uint32_t)u * ustride;
        len_next[u] = r < nb ? (int)*reinterpret_cast<const uint32_t*>(lbase + r * (uint32_t)(T * 4)) : 0;
    }
    if (lds) {
        for (int j = threadIdx.x; j < S; j += blockDim.x) sparse_slice[j] = 0;
        __syncthreads();
    }
    const float fresh = trace_merge(lp.trace, lp.rate, 0.0f, 1.0f);
    for (uint32_t rb = (uint32_t)wave * LPW; rb < nb; rb += (uint32_t)U * ustride) {       // (wave-uniform: rb is the wave's first learner of the batch)
        uint32_t key[U][REGS]; float val[U][REGS]; int len[U]; uint32_t nk[U]; float sc[U]; uint8_t fl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = rb + grp + (uint32_t)u * ustride;
            const uint32_t rr = r < nb ? r : 0u;
            len[u] = len_next[u];                                                    // (0 beyond the block's last learner)
            nk[u] = (uint32_t)*reinterpret_cast<const uint16_t*>(ktb + rr * 2u);             // (slice-relative, as the lists' keys)
            sc[u] = *reinterpret_cast<const float*>(tmb + rr * 4u);
            fl[u] = *reinterpret_cast<const uint8_t*>(flb + rr);
            const uint32_t ro = rr * (uint32_t)(kSparseCap * 4) + (uint32_t)gl * 4u;  // byte offset of the group's lane in the learner's row of values (keys: half)
#pragma unroll
            for (int e = 0; e < REGS; ++e) { key[u][e] = 0xffffffffu; val[u][e] = 0.0f; }
            // NR: the registers the longest of the wave's sub-lists reaches, as a compile-time constant of a straight-line body (a wave-uniform branch per
            // register and loop instead cost ~50 taken branches per batch with short lists)
            sparse_dispatch_regs<REGS, G>(wave_max_of_groups<G>(len[u]), [&](auto nr) {
                constexpr int NR = decltype(nr)::value;
#pragma unroll
                for (int e = 0; e < NR; ++e)
                    if (e * G + gl < len[u]) {
                        key[u][e] = (uint32_t)*reinterpret_cast<const uint16_t*>(kbase + (ro >> 1) + (uint32_t)(e * G * 2));
                        val[u][e] = *reinterpret_cast<const float*>(vbase + ro + (uint32_t)(e * G * 4));
                    }
            });
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = rb + grp + (uint32_t)(U + u) * ustride;
            len_next[u] = r < nb ? (int)*reinterpret_cast<const uint32_t*>(lbase + r * (uint32_t)(T * 4)) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rb + (uint32_t)u * ustride >= nb) break;                            // (wave-uniform: not even the wave's first group has a learner)
            const uint32_t r = rb + grp + (uint32_t)u * ustride;
            const bool ok = r < nb;
            const uint32_t ro = r * (uint32_t)(kSparseCap * 4) + (uint32_t)gl * 4u;   // (used where ok)
            // (the loaded lengths + 1: an append can open the next register; a cut or a reset only shortens)
            sparse_dispatch_regs<REGS, G>(wave_max_of_groups<G>(len[u]) + 1, [&](auto nr) {
                constexpr int NR = decltype(nr)::value;
                int ln = len[u];                                                    // (0 where !ok)
                if (fl[u] & 4) ln = 0;                                              // Watkins's cut
                unsigned long long hits = 0ull;
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    const bool live = e * G + gl < ln;
                    const bool hit = live && key[u][e] == nk[u];
                    hits |= __ballot(hit);
                    val[u][e] = trace_merge(lp.trace, lp.rate, live ? val[u][e] : 0.0f, hit ? 1.0f : 0.0f);
                }
                const bool need = ok && ((uint32_t)(hits >> (grp * G)) & ((1u << G) - 1u)) == 0u;  // the group's learner brings a key its sub-list does not hold
                const bool full = ln >= CAP;
                int slot = ln;
                if constexpr (NR == REGS) {
                    if (__ballot(need && full) != 0ull) {                           // some group evicts (its every register is live: NR == REGS)
                        unsigned long long best = ~0ull;
#pragma unroll
                        for (int e = 0; e < REGS; ++e) {
                            const unsigned long long cand = ((unsigned long long)(__builtin_bit_cast(uint32_t, val[u][e]) & 0x7fffffffu) << 32) | (uint32_t)(e * G + gl);
                            best = cand < best ? cand : best;
                        }
                        const int ev = (int)(uint32_t)group_min_u64<G>(best);
                        if (full) slot = ev;
                    }
                }
                if (need && !full) ln += 1;
#pragma unroll
                for (int e = 0; e < NR; ++e)
                    if (need && slot == e * G + gl) { key[u][e] = nk[u]; val[u][e] = fresh; }
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    if (e * G + gl >= ln) continue;
                    const unsigned long long q = fx_quantise(sc[u] * val[u][e], inv_lsb);
                    if (q == 0) continue;
                    if (lds) atomicAdd(reinterpret_cast<unsigned long long*>(&sparse_slice[key[u][e]]), q);
                    else fx_add(&dst[key[u][e]], q);
                }
                if (fl[u] & 1) ln = 0;                                              // trace.reset()
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    const int sl = e * G + gl;
                    if (sl < ln) {
                        *reinterpret_cast<float*>(vbase + ro + (uint32_t)(e * G * 4)) = val[u][e];
                        if (need && sl == slot) *reinterpret_cast<uint16_t*>(kbase_w + (ro >> 1) + (uint32_t)(e * G * 2)) = (uint16_t)key[u][e];      // (the one key of the sub-list this step can change)
                    }
                }
                if (gl == 0 && ok) *reinterpret_cast<uint32_t*>(lbase + r * (uint32_t)(T * 4)) = (uint32_t)ln;
            });
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
static __global__ void k_sparse_trace_get(SparseTrace st, int T, int slice, int64_t i, float* __restrict__ out /* zero-filled [F][A] */) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x, cap = kSparseCap / T;
    if (s < kSparseCap && s % cap < (int)st.len[i * T + s / cap]) out[(s / cap) * slice + (int)st.keys[i * (int64_t)kSparseCap + s]] = st.vals[i * (int64_t)kSparseCap + s];
}

}  // namespace rsrl
