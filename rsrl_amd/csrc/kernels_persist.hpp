// kernels_persist.hpp -- shared weights, dense basis: the WHOLE train call as ONE persistent launch (k_shared_persist).
//
// Synchronous mini-batch rule: every learner's TD error of batch-step t is taken against the same W_t,
// W_{t+1} = W_t + sum of the learners' terms, every learner then samples with W_{t+1}.  One launch per batch-step
// (k_shared_step, models.hpp) pays a kernel boundary per step: 8.4 us per batch-step for 131 072 learners, of which ~2 us are
// work.  Here the grid (one 512-learner block per CU, all co-resident) stays on the chip for the whole call; the learners' state,
// action, episode counter and phi(s') live in registers, W_t in LDS, and the only thing that crosses CUs is the mini-batch delta:
// ONE all-reduce of A*F 64-bit fixed-point numbers per batch-step.
//
// The all-reduce is a reduce-scatter + all-gather through DATA-TAGGED GRANULES (MI355X_MICROARCH.md, hand-off price list,
// "granules for latency"; cdna_hip_programming.md Guideline 16, form R2): an 8-byte word {tag12 | value52} written by one
// relaxed agent-scope (sc1, write-through) store and polled with relaxed agent-scope loads.  The tag travels with the value, so
// there is no flag, no counter, no fence and no atomic read-modify-write anywhere:
//   hop 1   block b stores its A*F partial sums, two per 16-byte store (each 8-byte half validates itself), into
//           A[pair][b][2]; block c (the OWNER of pair c) polls that pair's 2 x nb granules, one per thread, and adds them up --
//           integers: exact, any order;
//   hop 2   the owner stores the two totals into B[parity][rank][entry] of EVERY rank (its own included; over xGMI between
//           GPUs -- the one-hop peer-write of SURVEY 8e folded into the same kernel); thread e of every block polls entry e of
//           every rank and adds the ranks up (integers again: every replica gets the same bits).
// Measured on MI355X (scripts/ubench/granule_allreduce.hip, profiles/r03_ubench_granule_allreduce.txt), 256 blocks, per step:
// 3.2 us for this exchange, 6.1 us for device atomics + the guide's XCD-hierarchical barrier, 9.6-10.2 us for atomics + a flat
// counter barrier (round 2's 11.5).  Two fabric hops is the floor for an all-reduce with bounded fan-in.
//
// Hazards (single-buffered A, two parities of B): a block rewrites its A slots for step t+1 only after it has read every total
// of step t, i.e. after every owner has finished reading A of step t.  An owner publishes step t+1 only after all blocks OF ITS
// RANK have finished step t; blocks of another rank may still be reading step t -- hence the parity.  A rank cannot be two
// steps ahead of another (it needs the other's totals of the step in between).
// Tags: ((xs + 1) mod 4095) + 1 of the EXCHANGE SEQUENCE NUMBER xs (the number of batch-steps this ctx has pushed through these
// buffers: it only ever grows, whatever happens to the batch-step counter, and ranks in lock-step agree on it) -- never 0 (the
// cleared state), different for consecutive steps; every slot is rewritten every step (B: every second step), so a stale
// granule can never carry the wanted tag.  Every spin is bounded by the wall clock: a missing block or rank sets an
// error word, the step's update is NOT applied, the launch ends, and the next synchronising call reports it.
//
// The arithmetic is k_shared_step's, operation by operation (same projection, same LDS dot products, same MFMA rank-1 chain
// per wave, same quantisation of the block sums, same single rounding of the total): bit-identical to the one-launch-per-step
// path and to the oracle's orc_run_train_shared_dev (tests/test_gpu_bitwise.py).
#pragma once

#include "models.hpp"

namespace rsrl {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

struct PersistExch {
    unsigned long long* A;             // [pairs][nb][2] granules: this rank's hop-1 buffer
    unsigned long long* const* B;      // [world] pointers to every rank's hop-2 buffer [2][world][AFpad] (own rank included)
    unsigned long long* B_self;        // this rank's hop-2 buffer
    uint32_t* err;                     // != 0: a wait timed out (sticky)
    int world, rank;
    uint64_t timeout_ticks;            // of the 100 MHz wall clock
};

__device__ __forceinline__ unsigned long long g52(long long v, unsigned tag) {
    return ((unsigned long long)(tag & 0xfffu) << 52) | ((unsigned long long)v & 0xfffffffffffffull);
}
__device__ __forceinline__ long long g52_value(unsigned long long x) { return (long long)(x << 12) >> 12; }
__host__ __device__ __forceinline__ unsigned persist_tag(uint64_t t) { return (unsigned)((t + 1u) % 4095u) + 1u; }

// poll one granule until it carries `tag`; false (and *err set) when the wall clock runs out or an earlier wait already failed
template <bool SYSTEM>
__device__ __forceinline__ bool poll52(const unsigned long long* p, unsigned tag, uint32_t* err, uint64_t timeout_ticks, long long& out) {
    gu64* g = (gu64*)p;
    unsigned long long x = SYSTEM ? __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(x >> 52) != tag) {
        const uint64_t t_start = wall_clock64();
        unsigned spins = 0;
        do {
            if ((++spins & 63u) == 0u) {
                if (wall_clock64() - t_start > timeout_ticks || __hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    out = 0;
                    return false;
                }
            }
            x = SYSTEM ? __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned)(x >> 52) != tag);
    }
    out = g52_value(x);
    return true;
}

// n_steps batch-steps t0 .. t0 + n_steps - 1, then the closing phase C -- what train_now otherwise issues as n_steps + 1 launches
// of k_shared_step.  MULTI: the hop-2 stores / loads cross GPUs (system scope); world == 1 otherwise.
template <class M, int BLOCK, bool MULTI>
__global__ __launch_bounds__(BLOCK) void k_shared_persist(Common c, BasisGeom g, uint64_t t0, uint64_t xs0, int n_steps, float* __restrict__ W, PersistExch x,
                                                          DevStats* __restrict__ stats) {
    static_assert(M::kDense, "dense bases only");
    constexpr int D = M::D, A = M::A, F = M::F, AF = A * F, PAIRS = (AF + 1) / 2, AFP = 2 * PAIRS;
    constexpr int NWV = BLOCK / 64;
    static_assert(F <= 64 && A <= 4, "features across the 64 lanes, actions across the 4 rows of an MFMA block");
    const int64_t N = c.n_envs;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x, tid = (int)threadIdx.x;
    const int64_t i = (int64_t)b * BLOCK + tid;
    const int lane = tid & 63, wave = tid >> 6;
    const bool member = i < N;
    __shared__ __attribute__((aligned(16))) float sh_w[AF];
    __shared__ __attribute__((aligned(16))) float tile[NWV][F][kRank1Stride];
    __shared__ float part[NWV][AF];
    __shared__ unsigned long long red[2];
    __shared__ int sh_ok;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;

    // the learner: registers for the whole call
    const int64_t il = member ? i : N - 1;
    float s[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + il];
    uint32_t ep = c.ep_step[il];
    int a = c.action[il];
    const uint32_t gid = (uint32_t)(c.env_offset + il);
    const uint32_t cap = c.max_episode_steps;
    typename M::Feat fs, f0;
    { float s0[D]; M::Dom::reset(s0); M::features(s0, g, f0); }        // phi(s0): what every restarted episode projects
    M::features(s, g, fs);
    if (!member) {
#pragma unroll
        for (int f = 0; f < F; ++f) fs.phi[f] = 0.0f;
    }
    if (tid < AF) sh_w[tid] = W[tid];
    if (tid < 2) red[tid] = 0;
    if (tid == 0) sh_ok = 1;
    __syncthreads();
    const FxScale fx(c.alg.lr);                                        // the fixed-point resolution of the delta (models.hpp)
    // the draws of the next batch-step are generated in the shadow of the exchange: x_c = the behaviour policy's draw of phase C,
    // xin = the agent's own draw of phase A (SARSA)
    U4 x_c = U4{0, 0, 0, 0}, xin = U4{0, 0, 0, 0};
    if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, gid, t0, BLK_INNER);

    for (int j = 0; j <= n_steps; ++j) {
        const uint64_t t = t0 + (uint64_t)j;
        const bool do_c = j > 0, do_a = j < n_steps;
        float scale = 0.0f, delta = 0.0f, r = 0.0f;
        bool term = false, trunc = false;
        typename M::Feat fn;
        float ns[D];
        if (member) {
            float q_s[A];
            M::q_all_lds(sh_w, fs, q_s);
            if (do_c) a = policy_sample<A>(c.pol, q_s, x_c);            // ---- phase C of batch-step t-1: policy.sample with W_t
            if (do_a) {                                                 // ---- phase A of batch-step t
#pragma unroll
                for (int d = 0; d < D; ++d) ns[d] = s[d];
                term = M::Dom::step(ns, a, r);
                ep += 1;
                trunc = !term && cap > 0 && ep >= cap;
                M::features(ns, g, fn);
                float q_n[A];
                M::q_all_lds(sh_w, fn, q_n);
                float e;
                delta = td_dispatch<A>(c.alg, c.apol, q_s, a, q_n, r, term, xin, e);
                scale = c.alg.lr * e;
            }
        }
        if (!do_a) break;
        // ---- block-level sum of the learners' terms: k_shared_step's MFMA rank-1 chains, wave by wave (models.hpp)
        wave_rank1_sum<A, F>(tile[wave], lane, scale, fs.phi, member, a, part[wave]);
        __syncthreads();
        // ---- hop 1, publish: thread p stores the block's quantised sums of entries 2p, 2p+1 with one 16-byte sc1 store
        const uint64_t xs = xs0 + (uint64_t)j;                          // exchange sequence number: tags and parity (only ever grows)
        const unsigned tag = persist_tag(xs);
        if (tid < PAIRS) {
            long long q2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * tid + h;
                float tot = 0.0f;
                if (e < AF) {
                    tot = part[0][e];
#pragma unroll
                    for (int w = 1; w < NWV; ++w) tot += part[w][e];
                }
                q2[h] = (long long)fx_quantise(tot, fx.inv_lsb);
            }
            typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
            typedef int i4 __attribute__((ext_vector_type(4)));
            const u2 v = {g52(q2[0], tag), g52(q2[1], tag)};
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x.A, 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, (int)(((size_t)tid * nb + b) * 16), 0, 16 /* sc1: write-through */);
        }
        // ---- in the shadow of the exchange (the totals need two fabric hops, ~3 us): everything of the next batch-step that depends
        //      neither on W_{t+1} nor on the next action -- the episode restart, phi(s') as the next phi(s), the draws, the statistics
        if (member) {
            const bool done = term || trunc;
            sum_abs += (double)fabsf(delta); sum_r += (double)r;
            if (done) { n_ep += 1; n_trunc += trunc ? 1 : 0; sum_len += ep; }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
            fs = fn;                                                    // phi(s') is phi(s) of the next batch-step: projected once
            if (done) { M::Dom::reset(s); ep = 0; fs = f0; }
            x_c = draw(c.seed, gid, t, BLK_STEP);
            if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, gid, t + 1, BLK_INNER);
        }
        // ---- hop 1, reduce: this block owns pairs b, b + nb, ...; thread jj polls half jj & 1 of source block jj >> 1
        bool ok = true;
        for (int p = b; p < PAIRS; p += nb) {
            long long wv = 0;
            for (int jj = tid; jj < 2 * nb; jj += BLOCK) {
                long long v;
                ok &= poll52<false>(x.A + (size_t)p * 2 * nb + jj, tag, x.err, x.timeout_ticks, v);
                wv += v;
            }
            // even lanes hold entry 2p, odd lanes entry 2p+1 (BLOCK is even): wave sums, then one LDS atomic per wave and entry
#pragma unroll
            for (int o = 32; o > 1; o >>= 1) wv += __shfl_xor(wv, o, 64);
            if (lane < 2 && wv != 0) atomicAdd(&red[lane], (unsigned long long)wv);
            __syncthreads();
            if (tid < 2) {
                // hop 2, publish: this rank's total of entry 2p + tid into every rank's buffer
                const size_t slot = ((size_t)(xs & 1) * x.world + x.rank) * AFP + 2 * p + tid;
                const unsigned long long gv = g52((long long)red[tid], tag);
                if (MULTI) {
                    for (int rr = 0; rr < x.world; ++rr) __hip_atomic_store((gu64*)(x.B[rr] + slot), gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    __hip_atomic_store((gu64*)(x.B_self + slot), gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                red[tid] = 0;
            }
            __syncthreads();
        }
        // ---- hop 2, gather: thread e polls entry e of every rank and adds the ranks up; W_{t+1} = W_t + fl(total * lsb)
        float w_next = 0.0f;
        if (tid < AF) {
            long long total = 0;
            for (int rr = 0; rr < (MULTI ? x.world : 1); ++rr) {
                long long v;
                ok &= poll52<MULTI>(x.B_self + ((size_t)(xs & 1) * x.world + rr) * AFP + tid, tag, x.err, x.timeout_ticks, v);
                total += v;
            }
            w_next = sh_w[tid] + (float)total * fx.lsb;
        }
        if (!ok) sh_ok = 0;
        __syncthreads();
        if (!sh_ok) break;                                              // a wait timed out: the update is NOT applied, the launch ends
        if (tid < AF) sh_w[tid] = w_next;
        __syncthreads();
    }

    if (member) {
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.ep_step[i] = ep;
        c.action[i] = a;
    }
    if (b == 0 && tid < AF) W[tid] = sh_w[tid];
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

}  // namespace rsrl
