// rsrl_hip.hip -- C ABI (include/rsrl_hip.h) over the gfx950 kernels.
//
// One ctx = one HIP device + one stream + one (domain, basis, algo, policy, N, W-mode)
// instance, i.e. what the reference builds in examples/q_learning.rs:19-32.  There is no
// CPU path in this library: every entry point launches HIP kernels.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rsrl_hip.h"
#include "launch.hpp"
#include "models.hpp"
#include "model_list.hpp"
#include "kernels_wave.hpp"
#include "kernels_lambda.hpp"
#include "kernels_gq.hpp"
#include "kernels_td.hpp"
#include "kernels_qsigma.hpp"
#include "kernels_persist.hpp"
#include "kernels_wave_lambda.hpp"
#include "kernels_wave_aux.hpp"
#include "kernels_sparse_lambda.hpp"
#include "kernels_trait.hpp"

using namespace rsrl;

static inline bool is_lambda(int algo) { return algo == RSRL_SARSA_LAMBDA || algo == RSRL_Q_LAMBDA; }
static inline bool is_pred(int algo) { return algo == RSRL_TD || algo == RSRL_TD_LAMBDA; }        // one weight column (V function)
static inline bool has_aux(int algo) { return is_lambda(algo) || algo == RSRL_GREEDY_GQ || algo == RSRL_TD_LAMBDA; }   // second matrix of W's shape

namespace {
// tile coding, shared W: sum the n_rep copies of the FIXED-POINT delta table (and clear them) -- exact 64-bit integer sums,
// converted once: single rank W += fl(sum * lsb), otherwise that float goes to dW for the exchange.  n is a multiple of 2.
__global__ __launch_bounds__(256) void k_apply_rep(float* __restrict__ W, float* __restrict__ dW, long long* __restrict__ rep, int n_rep, int n, float lsb) {
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (j >= n) return;
    if (j + 1 >= n || (n & 1)) {                      // odd table sizes (copies not 16-byte aligned): one entry at a time
        for (int e = j; e < n && e < j + 2; ++e) {
            long long a = 0;
            for (int r = 0; r < n_rep; ++r) { long long* p = rep + (int64_t)r * n + e; const long long v = *p; if (v != 0) { a += v; *p = 0; } }
            const float d = (float)a * lsb;
            if (W) W[e] += d; else dW[e] = d;
        }
        return;
    }
    // all copies' loads go out together (one memory round trip instead of n_rep dependent ones), then the touched ones are cleared
    constexpr int kMaxRep = 16;
    longlong2 v[kMaxRep];
#pragma unroll
    for (int r = 0; r < kMaxRep; ++r)
        v[r] = r < n_rep ? *reinterpret_cast<const longlong2*>(rep + (int64_t)r * n + j) : make_longlong2(0, 0);
    long long a0 = 0, a1 = 0;
#pragma unroll
    for (int r = 0; r < kMaxRep; ++r) {
        if (r < n_rep && (v[r].x != 0 || v[r].y != 0)) {
            a0 += v[r].x; a1 += v[r].y;
            *reinterpret_cast<longlong2*>(rep + (int64_t)r * n + j) = make_longlong2(0, 0);
        }
    }
    const float d0 = (float)a0 * lsb, d1 = (float)a1 * lsb;
    if (W) { W[j] += d0; W[j + 1] += d1; }
    else { dW[j] = d0; dW[j + 1] = d1; }
}
// Shared tile coding, the scatter as a kernel of its own.  Block (chunk c, tiling t) takes the terms of `per_block` consecutive
// learners for ONE tiling: LDS slice of that tiling (64-bit fixed point), one LDS atomic per learner, ONE sweep, one device atomic
// per touched entry into copy c % n_rep of the table.  Against scattering inside the step kernel (1 024 learners x 8 tilings per
// block: a sweep per tiling per 1 024 learners, ~300 touched entries each) a block here covers 8x the learners per sweep and
// per flush: an eighth of the sweeps, a quarter of the device atomics.  The sums are integers: the same table whatever the
// grouping -- bit-identical to the fused scatter and to the oracle.
__global__ __launch_bounds__(1024) void k_tile_scatter(const uint16_t* __restrict__ keys, const float* __restrict__ terms, int64_t N, int S,
                                                       int per_block, long long* __restrict__ dW64, int n_rep, int64_t rep_stride, float inv_lsb) {
    extern __shared__ long long scatter_slice[];
    const int t = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * per_block;
    const int64_t i1 = i0 + per_block < N ? i0 + per_block : N;
    const uint16_t* __restrict__ kt = keys + (int64_t)t * N;
    float sc[8]; uint16_t kk[8];
    auto fetch = [&](int64_t ib) {                                       // eight learners per thread, their loads in flight together
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t i = ib + e * (int64_t)blockDim.x + threadIdx.x;
            sc[e] = i < i1 ? terms[i] : 0.0f;
            kk[e] = i < i1 ? kt[i] : (uint16_t)0;
        }
    };
    fetch(i0);                                                           // the first (usually the only) batch is on its way while the slice is cleared
    for (int j = threadIdx.x; j < S; j += blockDim.x) scatter_slice[j] = 0;
    __syncthreads();
    for (int64_t ib = i0; ib < i1; ib += 8 * (int64_t)blockDim.x) {
        if (ib != i0) fetch(ib);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned long long q = fx_quantise(sc[e], inv_lsb);      // the learner's term as ONE integer, the same for all its tilings
            if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&scatter_slice[kk[e]]), q);
        }
    }
    __syncthreads();
    long long* __restrict__ dst = dW64 + (int64_t)(blockIdx.x % (unsigned)n_rep) * rep_stride + (int64_t)t * S;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const long long v = scatter_slice[j];
        if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&dst[j]), (unsigned long long)v);
    }
}

// rsrl_hip_handle on shared weights: the mini-batch's fixed-point delta table -> float delta, table cleared
__global__ __launch_bounds__(256) void k_fx_finalize(long long* __restrict__ fx, float* __restrict__ dW, int n, float lsb) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long v = fx[j];
    dW[j] = (float)v * lsb;
    if (v != 0) fx[j] = 0;
}
// actions index weight columns: whatever a caller stored through a DEVICE pointer is brought into [0, A)
__global__ void k_clamp_actions(int32_t* __restrict__ a, int64_t n, int A) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = a[i]; a[i] = v < 0 ? 0 : (v > A - 1 ? A - 1 : v); }
}
// ---- RSRL_EXCHANGE_PEER: one-hop peer-write exchange of the shared-W delta ------------------------------
// Every rank stores its delta into slot [parity][rank] of EVERY rank's receive buffer (hipIpc-mapped: xGMI stores across
// GPUs), as naturally aligned 8-byte granules {value bits, tag = low 32 bits of (batch-step + 1)} written by one
// system-scope store each -- the tag travels with the value, so there is no separate flag, no fence and no second hop
// (MI355X_MICROARCH.md, hand-off price list: "granules for latency").  Each rank then sums the world slots in RANK order:
// every replica adds the same numbers in the same order => the replicas of W stay bit-identical, whatever the arrival
// order.  Two parities: a rank can be at most one exchange ahead of the slowest one (it cannot pass exchange t+1 before
// every peer has pushed t+1, i.e. finished reading t).
__global__ __launch_bounds__(256) void k_peer_push(const float* __restrict__ dW, int n, uint2* const* __restrict__ peers, int world, int rank,
                                                   uint64_t t, const uint64_t* __restrict__ t_dev, int64_t xdelta) {
    if (t_dev) t += *t_dev;
    const uint64_t xs = t + (uint64_t)xdelta;          // exchange sequence number: parity and tag (see Common::xdelta)
    // (a grid-stride loop: the grid is capped where several ranks share one device, peer_grid() below)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint64_t g = (uint64_t)__float_as_uint(dW[j]) | ((uint64_t)(uint32_t)(xs + 1) << 32);
        const size_t slot = ((size_t)(xs & 1) * world + rank) * (size_t)n + j;
        for (int r = 0; r < world; ++r)
            __hip_atomic_store(reinterpret_cast<uint64_t*>(peers[r] + slot), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// sum over ranks (ascending) of slot [parity][r][j], each polled until its tag says "exchange xs".  The spin is bounded by the
// wall clock (100 MHz; `timeout` ticks): a missing peer sets *err instead of hanging the GPU, and the sum is POISONED (NaN) --
// a partial sum is never applied silently.
__device__ __forceinline__ float peer_sum(const uint2* __restrict__ recv, int n, int world, int j, uint64_t xs, uint32_t* __restrict__ err, uint64_t timeout) {
    const uint32_t want = (uint32_t)(xs + 1);
    const uint64_t t_start = wall_clock64();
    bool failed = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // an earlier exchange timed out: fail fast, do not wait again
    float acc = 0.0f;
    for (int r = 0; r < world; ++r) {
        const uint64_t* p = reinterpret_cast<const uint64_t*>(recv + ((size_t)(xs & 1) * world + r) * (size_t)n + j);
        uint64_t g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while ((uint32_t)(g >> 32) != want) {
            if (failed || wall_clock64() - t_start > timeout) { atomicOr(err, 1u); failed = true; break; }
            __builtin_amdgcn_s_sleep(8);
            g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        acc += __uint_as_float((uint32_t)g);
    }
    return failed ? __builtin_nanf("") : acc;
}
__global__ __launch_bounds__(256) void k_peer_reduce(float* __restrict__ dW, int n, const uint2* __restrict__ recv, int world, uint64_t t,
                                                     const uint64_t* __restrict__ t_dev, int64_t xdelta, uint32_t* __restrict__ err, uint64_t timeout) {
    if (t_dev) t += *t_dev;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        dW[j] = peer_sum(recv, n, world, j, t + (uint64_t)xdelta, err, timeout);
}
// multi-rank mode: the fold as a kernel of its own (the copies of batch-step t's fixed-point delta table -> one float per
// output), feeding the exchange
__device__ __forceinline__ float tab_total(const long long* __restrict__ tab, int n, int j, float lr, uint64_t t) {
    DeltaTab dt(const_cast<long long*>(tab), n, lr, t);
    long long s = 0;
#pragma unroll
    for (int r = 0; r < kTabRep; ++r) s += dt.out[r * n + j];
    return (float)s * dt.lsb;
}
__global__ __launch_bounds__(kBlock) void k_tab_finalize(const long long* __restrict__ tab, int n, float lr, float* __restrict__ dW, uint64_t t,
                                                         const uint64_t* __restrict__ t_dev) {
    if (t_dev) t += *t_dev;
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j < n) dW[j] = tab_total(tab, n, j, lr, t);
}

// dense peer path, ONE launch instead of four: the delta of batch-step t goes from the fixed-point table straight into every
// rank's receive slot, and the same thread then sums the ranks' slots (rank order, bounded wait as k_peer_reduce) into W.
// Every rank pushes before it waits, so the ranks cannot wait for each other's pushes in a cycle.
__global__ __launch_bounds__(256) void k_tab_exchange_apply(const long long* __restrict__ tab, int n, float lr, uint2* const* __restrict__ peers,
                                                            const uint2* __restrict__ recv, float* __restrict__ W, int world, int rank, uint64_t t,
                                                            const uint64_t* __restrict__ t_dev, int64_t xdelta, uint32_t* __restrict__ err, uint64_t timeout) {
    if (t_dev) t += *t_dev;
    const uint64_t xs = t + (uint64_t)xdelta;
    // every element is pushed BEFORE the first wait (two grid-stride loops: the grid may be capped, peer_grid() below)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float tot = tab_total(tab, n, j, lr, t);
        const uint64_t mine = (uint64_t)__float_as_uint(tot) | ((uint64_t)(uint32_t)(xs + 1) << 32);
        const size_t slot = ((size_t)(xs & 1) * world + rank) * (size_t)n + j;
        for (int r = 0; r < world; ++r)
            __hip_atomic_store(reinterpret_cast<uint64_t*>(peers[r] + slot), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        W[j] += peer_sum(recv, n, world, j, xs, err, timeout);
}
__global__ void k_fill_f32(float* __restrict__ p, int64_t n, float v) { const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_set_dyn(DynParams* __restrict__ d, DynParams v) { *d = v; }
__global__ void k_set_t(uint64_t* __restrict__ t_dev, uint64_t v) { *t_dev = v; }
__global__ void k_advance_t(uint64_t* __restrict__ t_dev, uint64_t d) { *t_dev += d; }
__global__ void k_apply_dw(float* __restrict__ W, float* __restrict__ dW, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) { W[j] += dW[j]; dW[j] = 0.0f; }
}

// get/set of one learner's weights as row-major f32[F][A] (ndarray (F, A))   params/mod.rs:116-134
// device layouts: Fourier W[A][F][Nw] (learner fastest); tile coding W[Nw][F][A]
__device__ __forceinline__ int64_t w_index(bool tile, int64_t stride, int64_t wi, int F, int A, int f, int b) {
    return tile ? (wi * (int64_t)F + f) * A + b : ((int64_t)(b * F + f)) * stride + wi;
}
__global__ void k_weights_get(const float* __restrict__ W, bool tile, int64_t stride, int64_t wi, int F, int A, float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F * A) return;
    out[j] = W[w_index(tile, stride, wi, F, A, j / A, j % A)];
}
__global__ void k_weights_set(float* __restrict__ W, bool tile, int64_t stride, int64_t wi, int F, int A, const float* __restrict__ in) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F * A) return;
    W[w_index(tile, stride, wi, F, A, j / A, j % A)] = in[j];
}
// grid.x covers the learners, grid.y strides over the F*A weights
__global__ void k_weights_set_all(float* __restrict__ W, bool tile, int64_t N, int64_t stride, int64_t ls, int F, int A, const float* __restrict__ in) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    for (int j = blockIdx.y; j < F * A; j += gridDim.y) W[w_index(tile, stride, i * ls, F, A, j / A, j % A)] = in[j];
}
// the same sum for a learner-major W[N][AF], with every word weighted by the index it has in the feature-major layout
// ((row)*N + learner): the checksum of the weights does not depend on the layout the ctx chose
__global__ void k_checksum_lm(const uint32_t* __restrict__ p, int64_t N, int AF, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    const size_t n = (size_t)N * AF;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const size_t learner = j / AF, row = j % AF;
        acc += (unsigned long long)p[j] * (2ull * (row * (size_t)N + learner) + 1ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
// order-independent checksum: sum over words of bits * (2*index + 1)  (mod 2^64)
__global__ void k_checksum(const uint32_t* __restrict__ p, size_t n, size_t index_offset, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
        acc += (unsigned long long)p[j] * (2ull * (j + index_offset) + 1ull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
}  // namespace

// ------------------------------------------------------------------------------- errors
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_last_error = buf;
    // the HIP runtime keeps the last failure of ANY call of this thread until somebody asks for it: asked for here WHEN THE FAILURE REPORTED IS A HIP ONE, so
    // that it is not found again by the next launch check (KCHECK) of a healthy ctx.  A pure argument / state error (EINVAL, ESTATE) leaves it alone: a launch
    // error not yet checked must not be lost behind an unrelated report; rsrl_hip_destroy and the clean-up paths clear it themselves.
    if (code == RSRL_HIP_EHIP || code == RSRL_HIP_ENOMEM) (void)hipGetLastError();
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(_e == hipErrorOutOfMemory ? RSRL_HIP_ENOMEM : RSRL_HIP_EHIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(_e), __FILE__, __LINE__);                         \
    } while (0)
#define CHECK_CTX(ctx) do { if (!(ctx)) return fail(RSRL_HIP_EINVAL, "null ctx"); } while (0)

// ------------------------------------------------------------------------------- ctx
struct Scratch { void* p = nullptr; size_t cap = 0; };

struct rsrl_hip_ctx {
    rsrl_hip_config cfg{};
    int D = 0, A = 0, F = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    float* state = nullptr; int32_t* action = nullptr; uint32_t* ep_step = nullptr;
    float* W = nullptr; float* dW = nullptr;
    int Aw = 0;                      // columns of the weight matrix: A (control) or 1 (prediction: ScalarLFA)
    long long* dW_rep = nullptr; int n_rep = 1;  // shared tile coding: n_rep copies of the fixed-point (64-bit) delta table
    long long* sh_tab = nullptr;     // shared-W dense basis: 3 sets x kTabRep copies of the fixed-point delta table (models.hpp DeltaTab)
    long long* h_fx = nullptr;       // shared W: fixed-point delta table of rsrl_hip_handle (one entry per weight)
    bool tile_slice = false;         // shared tile coding: one tiling's slice (twice, as 64-bit words) fits LDS
    uint16_t* sc_keys = nullptr;     // shared tile coding, separate scatter kernel: slice-relative entries [T][N]
    float* sc_terms = nullptr;       //   and terms lr*e [N] handed from the step kernel to k_tile_scatter
    uint64_t sh_tab_t = 0;           // batch-step counter the table rotation is in phase with (the end of the last shared train call)
    float* W2 = nullptr;             // shared-W dense basis: second weight buffer (k_shared_step reads one, block 0 writes the other)
    int sh_par = 0;                  // which W buffer holds the current weights (0 = W)
    unsigned sh_rows = 0;            // rows per buffer = blocks of k_shared_step
    float* qcache = nullptr;         // [A][N]: Q(s,.) carried between train launches (register family)
    float* qs_buf = nullptr; uint32_t* qs_head = nullptr; uint32_t* qs_len = nullptr;     // QSigma: per-learner n-step backups
    float* eps = nullptr;            // [N] per-learner EpsilonGreedy.epsilon (config.epsilon_decay != 1), else null
    // lambda agents over ONE shared tile table: every learner's sparse trace + the step's mailbox (kernels_sparse_lambda.hpp)
    uint32_t* sp_keys = nullptr; float* sp_vals = nullptr; uint32_t* sp_len = nullptr;    // sparse traces: [N][kSparseCap] x 2, lengths [N][n_tilings]
    bool sp_lds = false;             //   one tiling's slice of the delta table fits LDS (k_sparse_trace_scatter)
    float* Z = nullptr;              // auxiliary matrix f32[A][F][N]: eligibility traces (lambda agents) / fa_td weights (GreedyGQ)
    bool q_valid = false;            // false whenever weights / states were changed from outside the driver loop
    // ---- the trait-granular fast path (kernels_trait.hpp): register-family Fourier basis, per-learner f32 weights, learner-major layout
    float* tq_key = nullptr;         // [D][N]: the state each learner's qcache entry belongs to (allocated iff the ctx takes the fast path)
    bool tq_valid = false;           // qcache / tq_key hold the hand-over of rsrl_hip_handle (false: the keys are emptied before the next trait kernel)
    // calls of the trait-granular loop accepted but not launched yet (ctx-owned stream, device pointers, the loop's own order):
    //   stage 1 = domain_step, 2 = + handle on exactly that transition, 3 = + domain_reset with the terminal flags as its mask;
    //   policy_sample(NULL) then launches the whole batch-step as ONE kernel; anything else launches the accepted calls one by one first
    struct TraitPend { int stage = 0; const int32_t* act = nullptr; float* from = nullptr; float* to = nullptr; float* rew = nullptr;
                       uint8_t* term = nullptr; float* td = nullptr; uint64_t t_handle = 0; } tp;
    uint8_t* flags = nullptr;        // shared-W: terminal/truncated flags between phase A and phase C
    size_t w_elems = 0; size_t dw_elems = 0; size_t w_bytes = 0; size_t z_bytes = 0;      // (the auxiliary matrix Z -- traces / fa_td weights -- is f32 whatever W's storage)
    int64_t w_stride = 0;            // stride between (action, feature) rows of W
    int64_t w_ls = 1;                // stride between learners (A*F in the learner-major single-step layout, else 1)
    DevStats* d_stats = nullptr; DevStats* h_stats = nullptr;   // one slot per thread block
    size_t n_stat_slots = 0;
    bool k1_quad = false;                      // single-step streaming kernel with four lanes per learner (k_step_reg_q4)
    uint64_t t = 0;          // batch-steps executed (RNG counter)
    int64_t pending = 0;     // batch-steps accepted by rsrl_hip_train but not launched yet (launch coalescing, see rsrl_hip_train)
    uint64_t api_calls = 0;  // RNG counter of rsrl_hip_policy_sample
    uint64_t rollout_calls = 0;   // ... and of rsrl_hip_rollout_policy (one stream of draws per call)
    Scratch scratch[8];
    // timing of train launches
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<uint32_t> event_launches;      // batch-step launches bracketed by each event pair (a graph replay brackets many)
    size_t events_used = 0;
    // launch-bound inner loops (one batch-step per launch: the streaming kernel, the shared-W phases) replayed as a hipGraph
    uint64_t* d_t = nullptr;                   // device copy of the batch-step counter: graph nodes carry offsets to it
    DynParams* d_dyn = nullptr;                // device copy of the policy parameters graph nodes read (epsilon can change between calls)
    DynParams dyn_uploaded{};                  // what d_dyn holds
    bool dyn_valid = false;
    hipGraph_t step_graph = nullptr;
    hipGraphExec_t step_graph_exec = nullptr;
    Common step_graph_key{};                   // kernel arguments the graph was captured with
    int step_graph_kind = 0;                   // 1 = k_step_reg, 2 = shared-W batch-step
    const char* kernel_name = "";
    // multi-rank shared-W (one process per GPU): the per-batch-step exchange of the weight delta
    ncclComm_t comm = nullptr;                 // RSRL_EXCHANGE_RCCL
    int n_simd = 1024;                         // SIMDs of the device (4 per CU): launches of more waves than that co-schedule waves
    int world_size = 1, rank = 0;
    bool multi = false;                        // an exchange is attached (a communicator of size 1 included: same sequence)
    // RSRL_EXCHANGE_PEER: one-hop peer-write.  recv = this rank's receive buffer, granules {value bits, step tag}
    // [2 (step parity)][world][dw_elems]; peers[r] = rank r's receive buffer mapped into this process (hipIpc), own included
    uint2* peer_recv = nullptr; size_t peer_recv_bytes = 0; int peer_world = 0;
    std::vector<void*> peer_ptrs; std::vector<char> peer_opened;
    uint2** d_peer_ptrs = nullptr;             // device copy of peer_ptrs
    uint32_t* d_peer_err = nullptr;            // set by a block / rank that waited too long for a peer (sticky; shared-W ctxs only)
    uint64_t peer_seq = 0;                     // exchanges performed on peer_recv so far: parity and tags follow it (Common::xdelta)
    uint64_t peer_timeout = 400000000ull;      // bound of every in-kernel wait, ticks of the 100 MHz wall clock (RSRL_PEER_TIMEOUT_MS, default 4000)
    size_t peer_old_bytes = 0;                 // peer_recv = [granules of the per-step exchange kernels | hop-2 buffer of the persistent kernel]
    // persistent shared-W kernel (kernels_persist.hpp): hop-1 buffer A, hop-2 buffer B (own; inside peer_recv in peer mode)
    unsigned long long* px_A = nullptr; unsigned long long* px_B = nullptr; bool px_B_owned = false;
    unsigned long long** d_px_Bptrs = nullptr; // device array [world] of every rank's hop-2 buffer
    uint64_t px_seq = 0;                       // batch-steps exchanged through px_A / px_B so far (tags and parity)
    int n_cu = 256;
    // ---- co-residency of the persistent kernel (every block of the grid -- and of every peer rank -- must be resident at once)
    int persist_occ = -1;                      // blocks of k_shared_persist one CU admits (occupancy query; -1 = not asked yet, 0 = none)
    bool group_persist = false;                // PEER group: the COLLECTIVE decision of rsrl_hip_peer_connect (every rank takes the same path)
    int peer_share = 1;                        // ranks of this ctx's group on ITS device, itself included (rsrl_hip_peer_connect); caps the exchange grids
    bool coop_allowed = true;                  // no rank of this ctx's group shares (process, device) with it: a cooperative launch cannot queue behind a peer's
    bool coop_validated = false;               // one cooperative launch of this ctx's persistent grid has been accepted by the runtime
    bool persist_refused = false;              // ... or refused (single rank: the per-step path takes over for good)
    uint64_t group_token = 0;                  // identifies the peer group (same on every rank); 0 = a lone ctx
    bool st_rccl_group = false;                // member of a single-thread RCCL group of more than one rank: stepped by rsrl_hip_group_train only
};

static Common make_common(const rsrl_hip_ctx* c) {
    Common k{};
    k.n_envs = c->cfg.n_envs; k.env_offset = c->cfg.env_offset; k.seed = c->cfg.seed;
    k.pol.kind = c->cfg.policy;
    double v = c->cfg.epsilon * 16777216.0;
    k.pol.eps_thr = v <= 0.0 ? 0u : (v >= 16777216.0 ? 16777216u : (uint32_t)v);
    k.pol.eps = (float)c->cfg.epsilon; k.pol.tau = (float)c->cfg.tau;
    if (c->cfg.agent_policy < 0) { k.apol = k.pol; k.apol_same = 1; }
    else {
        k.apol.kind = c->cfg.agent_policy;
        v = c->cfg.agent_epsilon * 16777216.0;
        k.apol.eps_thr = v <= 0.0 ? 0u : (v >= 16777216.0 ? 16777216u : (uint32_t)v);
        k.apol.eps = (float)c->cfg.agent_epsilon; k.apol.tau = (float)c->cfg.agent_tau;
        k.apol_same = 0;
    }
    k.alg.kind = c->cfg.algo; k.alg.gamma = (float)c->cfg.gamma; k.alg.lr = (float)c->cfg.lr;
    k.alg.alpha = (float)c->cfg.alpha;
    k.max_episode_steps = c->cfg.max_episode_steps;
    k.state = c->state; k.action = c->action; k.ep_step = c->ep_step; k.W = c->W; k.w_stride = c->w_stride; k.w_ls = c->w_ls; k.shared = c->cfg.weight_mode == RSRL_W_SHARED ? 1 : 0;
    k.qcache = c->qcache; k.q_valid = c->q_valid ? 1 : 0;
    k.eps = c->eps; k.eps_decay = (float)c->cfg.epsilon_decay; k.eps_min = (float)c->cfg.epsilon_min;
    k.xdelta = (int64_t)c->peer_seq - (int64_t)c->t;
    return k;
}

static LambdaParams make_lambda(const rsrl_hip_ctx* c) {
    LambdaParams lp{};
    lp.Z = c->Z;
    double rate = c->cfg.gamma * c->cfg.lambda;
    if (c->cfg.trace == RSRL_TRACE_DUTCH) rate *= (1.0 - c->cfg.alpha);       // traces.rs:233-239
    lp.rate = (float)rate; lp.alpha = (float)c->cfg.alpha; lp.trace = c->cfg.trace;
    return lp;
}

static GqParams make_gq(const rsrl_hip_ctx* c) {
    GqParams gp{};
    gp.V = c->Z; gp.lr_td = (float)c->cfg.lr_td;
    return gp;
}

static QsParams make_qs(const rsrl_hip_ctx* c) {
    QsParams qp{};
    qp.buf = c->qs_buf; qp.head = c->qs_head; qp.len = c->qs_len; qp.n_steps = c->cfg.n_steps;
    qp.sigma = (float)c->cfg.sigma; qp.alpha = (float)c->cfg.alpha;
    return qp;
}

static TdParams make_td(const rsrl_hip_ctx* c) {
    TdParams tp{};
    tp.Z = c->Z;
    double rate = c->cfg.gamma * c->cfg.lambda;
    if (c->cfg.trace == RSRL_TRACE_DUTCH) rate *= (1.0 - c->cfg.alpha);
    tp.rate = (float)rate; tp.trace = c->cfg.trace;
    return tp;
}

// SARSALambda / QLambda over one shared tile-coded table: per-learner SPARSE traces (kernels_sparse_lambda.hpp)
static inline bool is_sparse_lambda(const rsrl_hip_config& cfg) {
    return is_lambda(cfg.algo) && cfg.basis == RSRL_TILE_CODING && cfg.weight_mode == RSRL_W_SHARED;
}
// GreedyGQ / TD / TDLambda on the order-7 wave family (kernels_wave_aux.hpp)
static inline bool is_wave_aux_algo(int algo) { return algo == RSRL_GREEDY_GQ || is_pred(algo); }
static WaveAuxParams make_wave_aux(const rsrl_hip_ctx* c) {
    WaveAuxParams ap{};
    ap.mode = c->cfg.algo == RSRL_GREEDY_GQ ? WAUX_GQ : (c->cfg.algo == RSRL_TD ? WAUX_TD : WAUX_TDL);
    ap.aux = c->Z; ap.lr_td = (float)c->cfg.lr_td;
    const TdParams tp = make_td(c);
    ap.rate = tp.rate; ap.trace = tp.trace;
    return ap;
}
template <int DM, class WT> struct WaveTag { static constexpr int domain = DM; using wt = WT; };
template <class Fn>
static bool for_wave(const rsrl_hip_ctx* c, Fn&& fn) {
    const bool bf = c->cfg.weight_dtype == RSRL_W_BF16;
    if (c->cfg.domain == RSRL_CART_POLE) { if (bf) fn(WaveTag<1, bf16_t>{}); else fn(WaveTag<1, float>{}); return true; }
    if (c->cfg.domain == RSRL_ACROBOT) { if (bf) fn(WaveTag<2, bf16_t>{}); else fn(WaveTag<2, float>{}); return true; }
    return false;
}
template <class... Args>
static void launch_wave_aux(const rsrl_hip_ctx* c, dim3 grid, const Common& k, const WaveAuxParams& ap, Args... args) {      // (W: the ctx's, in its storage type)
    for_wave(c, [&](auto tag) {
        using T = decltype(tag); using WT = typename T::wt;
        hipLaunchKernelGGL((k_wave_aux<T::domain, WT>), grid, dim3(kBlock), 0, c->stream, k, ap, (WT*)c->W, args...);
    });
}

// the step kernel of the shared-weight loops (what rsrl_hip_timing_read names): the dense bases', shared tile coding's, the sparse-trace lambda agents'
static inline const char* shared_kernel_name(const rsrl_hip_ctx* c) {
    return c->cfg.basis == RSRL_FOURIER ? "k_shared_step" : (c->sp_keys ? "k_sparse_trace_scatter" : "k_shared_ca");
}
static inline unsigned grid_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
// resolution of the fixed-point delta tables of shared tile coding: 2^(floor(log2 |lr|) - 28), the same bits the kernel derives
static inline float tile_lsb(float lr) {
    uint32_t u; memcpy(&u, &lr, 4);
    const uint32_t eb = (u >> 23) & 0xffu;
    const uint32_t ex = (eb < 30u ? 30u : eb) - 28u;
    const uint32_t v = ex << 23; float f; memcpy(&f, &v, 4);
    return f;
}
constexpr int kSharedBlock = 512;    // learners per block of k_shared_step: 256 blocks = one per CU for a 131 072-env shard

// ---- (basis, domain, parameter) -> Model type: model_list.hpp
static bool is_generic_fourier(const rsrl_hip_config& cfg) {
    if (cfg.basis != RSRL_FOURIER) return false;
#define X(TYPE, BS, DM, P) if (P != -1 && model_match(cfg, BS, DM, P)) return false;
    RSRL_MODELS(X)
#undef X
    return true;
}
static bool model_supported(const rsrl_hip_config& cfg) {
#define X(TYPE, BS, DM, P) if (model_match(cfg, BS, DM, P)) return true;
    RSRL_MODELS(X)
#undef X
    return false;
}
// calls fn(Tag<Model>{}) for the ctx's model; false if none matches
template <class Fn>
static bool for_model(const rsrl_hip_ctx* c, Fn&& fn) {
#define X(TYPE, BS, DM, P) if (model_match(c->cfg, BS, DM, P)) { fn(Tag<RSRL_UNPAREN TYPE>{}); return true; }
    RSRL_MODELS(X)
#undef X
    return false;
}
static BasisGeom make_geom(const rsrl_hip_ctx* c) {
    return BasisGeom{c->F, c->cfg.basis == RSRL_FOURIER ? c->cfg.order : c->cfg.tiles_per_dim};
}
// wave family (one wavefront per learner): Fourier order 7 on the 4-D domains, f32 or bf16 weights
static bool is_wave(const rsrl_hip_config& cfg) {
    return cfg.basis == RSRL_FOURIER && cfg.order == kWaveOrder && (cfg.domain == RSRL_CART_POLE || cfg.domain == RSRL_ACROBOT);
}
static inline unsigned wave_grid_for(int64_t items) { return (unsigned)((items + (kBlock / 64) - 1) / (kBlock / 64)); }
#define NO_MODEL(c) fail(RSRL_HIP_EINVAL, "no kernel for basis %d domain %d order %d tilings %d", (c)->cfg.basis, (c)->cfg.domain, (c)->cfg.order, (c)->cfg.n_tilings)

// ---- host/device pointer staging ---------------------------------------------------------
static bool is_device_ptr(const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
static int scratch_reserve(rsrl_hip_ctx* c, int slot, size_t bytes) {
    Scratch& s = c->scratch[slot];
    if (s.cap >= bytes) return RSRL_HIP_OK;
    if (s.p) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(s.p)); s.p = nullptr; s.cap = 0; }
    HIP_TRY(hipMalloc(&s.p, bytes));
    s.cap = bytes;
    return RSRL_HIP_OK;
}
// input: returns a device pointer holding the caller's data
template <class T>
static int stage_in(rsrl_hip_ctx* c, int slot, const T* user, size_t count, const T** dev) {
    if (!user) { *dev = nullptr; return RSRL_HIP_OK; }
    if (is_device_ptr(user)) { *dev = user; return RSRL_HIP_OK; }
    int rc = scratch_reserve(c, slot, count * sizeof(T));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->scratch[slot].p, user, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dev = (const T*)c->scratch[slot].p;
    return RSRL_HIP_OK;
}
// output: returns the device pointer kernels should write; flush copies back if user is host memory
template <class T>
struct OutBuf { T* user = nullptr; T* dev = nullptr; size_t count = 0; bool staged = false; };
template <class T>
static int stage_out(rsrl_hip_ctx* c, int slot, T* user, size_t count, OutBuf<T>* ob) {
    ob->user = user; ob->count = count; ob->staged = false; ob->dev = nullptr;
    if (!user) return RSRL_HIP_OK;
    if (is_device_ptr(user)) { ob->dev = user; return RSRL_HIP_OK; }
    int rc = scratch_reserve(c, slot, count * sizeof(T));
    if (rc) return rc;
    ob->dev = (T*)c->scratch[slot].p; ob->staged = true;
    return RSRL_HIP_OK;
}
template <class T>
static int flush_out(rsrl_hip_ctx* c, OutBuf<T>* ob, bool* need_sync) {
    if (ob->staged) {
        HIP_TRY(hipMemcpyAsync(ob->user, ob->dev, ob->count * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        *need_sync = true;
    }
    return RSRL_HIP_OK;
}
// caller-supplied actions index weight columns (W[:,a]): a HOST array is validated (EINVAL, where the reference would
// panic on the out-of-range column); a DEVICE array cannot be inspected from here and is clamped by the kernels instead
static int check_host_actions(const int32_t* a, size_t n, int A) {
    if (!a || is_device_ptr(a)) return RSRL_HIP_OK;
    for (size_t i = 0; i < n; ++i)
        if (a[i] < 0 || a[i] >= A) return fail(RSRL_HIP_EINVAL, "action[%zu] = %d is outside [0, %d)", i, a[i], A);
    return RSRL_HIP_OK;
}
// caller-supplied STATES: the reference's wrap! (rsrl_domains/src/macros.rs:14-24) brings an angle home by repeated +-2 pi -- a loop that does not end
// for an infinite value and practically not for a huge one (Acrobot; on the device that is a hung GPU).  A HOST array is validated (EINVAL: every
// component finite and within 1000 widths of its dimension's bounds); a DEVICE array cannot be inspected from here and is clamped into that range
// by k_clamp_states (NaN stays NaN: comparisons with it are false, nothing loops).
static void state_limits(const rsrl_hip_ctx* c, float* lo, float* hi);
static int check_host_states(const rsrl_hip_ctx* c, const float* s, size_t n_cols) {
    if (!s || is_device_ptr(s)) return RSRL_HIP_OK;
    float lo[8], hi[8];
    state_limits(c, lo, hi);
    for (int d = 0; d < c->D; ++d)
        for (size_t i = 0; i < n_cols; ++i) {
            const float x = s[(size_t)d * n_cols + i];
            if (!(x >= lo[d] && x <= hi[d]))
                return fail(RSRL_HIP_EINVAL, "state[%d][%zu] = %g is not a finite value within 1000 widths of the dimension's bounds [%g, %g]", d, i, (double)x,
                            (double)lo[d], (double)hi[d]);
        }
    return RSRL_HIP_OK;
}
struct StateLimits { float lo[8], hi[8]; };
// a DEVICE array of states is validated on the device, with the host path's rule (it used to be clamped silently, and NaN passed): *bad counts the
// components that are not finite values within the limits; the caller copies the array into the ctx only when there are none
__global__ void k_check_states(const float* __restrict__ s, int64_t n, int D, StateLimits lim, unsigned* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned cnt = 0;
    for (int d = 0; d < D; ++d) {
        const float x = s[(int64_t)d * n + i];
        cnt += (x >= lim.lo[d] && x <= lim.hi[d]) ? 0u : 1u;
    }
    if (cnt) atomicAdd(bad, cnt);
}
#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#define KCHECK() HIP_TRY(hipGetLastError())

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) return fail(RSRL_HIP_ERCCL, "%s failed: %s", #expr, ncclGetErrorString(_r)); \
    } while (0)
// The one exchange step of the path: sum the (F x A) f32 weight delta over the ranks so that every rank applies the
// identical update and the replicas of W stay bit-identical.  In place on c->dW, on the ctx's stream, no host
// synchronisation, capturable into the step graph.  A communicator of size 1 runs the same sequence (that is how the
// multi-rank path is exercised on a one-GPU box).  At 432 B (MountainCar Fourier(5)) this is latency-bound, not link-bound.
//   t / t_dev: the batch-step this exchange belongs to (PEER: slot parity and granule tag); t_dev != nullptr inside a graph.
// Grid of the peer-exchange kernels (grid-stride loops over the n outputs).  A rank that has its device to itself takes one block per 256 outputs, as
// before.  Ranks that SHARE a device (oversubscribed tests, several ranks per GPU) wait -- bounded -- for each other's pushes while occupying compute
// units: eight ranks x 768 waiting blocks (a 16 x 8^4 x 3 tile table) fill the device and the rank they wait for never gets a unit (found by
// tests/fuzz_ranks.py as an exchange time-out).  So the group's waiting blocks together may take at most HALF of the device's resident blocks.
static unsigned peer_grid(const rsrl_hip_ctx* c, int n) {
    const unsigned full = (unsigned)((n + 255) / 256);
    if (c->peer_share <= 1) return full;
    const unsigned resident = (unsigned)(c->n_cu > 0 ? c->n_cu : 256) * 8u;            // 256-thread blocks per device at full occupancy
    const unsigned cap = std::max(1u, resident / 2u / (unsigned)c->peer_share);
    return std::min(full, cap);
}
static int exchange_dw(rsrl_hip_ctx* c, uint64_t t, const uint64_t* t_dev, int64_t xdelta) {
    if (!c->multi) return RSRL_HIP_OK;
    const int n = (int)c->dw_elems;
    if (c->cfg.exchange == RSRL_EXCHANGE_PEER) {
        hipLaunchKernelGGL(k_peer_push, dim3(peer_grid(c, n)), dim3(256), 0, c->stream, c->dW, n, c->d_peer_ptrs, c->world_size, c->rank, t, t_dev, xdelta);
        hipLaunchKernelGGL(k_peer_reduce, dim3(peer_grid(c, n)), dim3(256), 0, c->stream, c->dW, n, c->peer_recv, c->world_size, t, t_dev, xdelta, c->d_peer_err,
                           c->peer_timeout);
        KCHECK();
        return RSRL_HIP_OK;
    }
    NCCL_TRY(ncclAllReduce(c->dW, c->dW, c->dw_elems, ncclFloat, ncclSum, c->comm, c->stream));
    return RSRL_HIP_OK;
}
// a rank that waited too long for a peer left a mark: report it at the next synchronising call
static int peer_check(rsrl_hip_ctx* c) {
    if (!c->d_peer_err) return RSRL_HIP_OK;
    uint32_t e = 0;
    HIP_TRY(hipMemcpyAsync(&e, c->d_peer_err, sizeof(e), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (e) return fail(RSRL_HIP_ERCCL, "shared-W exchange timed out: a block or rank did not deliver its weight delta (rank %d of %d); the update was not applied "
                                       "(per-step exchange: the weights are poisoned with NaN)", c->rank, c->world_size);
    return RSRL_HIP_OK;
}

static int flush_pending(rsrl_hip_ctx* c);      // launch what the ctx has accepted but not launched yet (train's coalesced batch-steps, deferred trait calls)
static int trait_flush(rsrl_hip_ctx* c);        // ... the deferred trait calls alone, one kernel per call
static int timing_begin(rsrl_hip_ctx* c);
static int timing_end(rsrl_hip_ctx* c, uint32_t launches = 1);
static int trait_cache_ready(rsrl_hip_ctx* c);
static inline bool trait_fast(const rsrl_hip_ctx* c);
static int launch_trait_handle(rsrl_hip_ctx* c, const Common& k, const float* from, const int32_t* act, const float* rew, const float* to,
                               const uint8_t* term, int64_t M, uint64_t t, float* td);
static int launch_domain_step(rsrl_hip_ctx* c, const Common& k, const int32_t* d_act, float* from, float* next, float* rew, uint8_t* term);
static int launch_domain_reset(rsrl_hip_ctx* c, const Common& k, const uint8_t* d_mask);
#define FLUSH(c) TRY(flush_pending(c))

// ------------------------------------------------------------------------------- API
extern "C" {

int rsrl_hip_abi_version(void) { return RSRL_HIP_ABI_VERSION; }
int rsrl_hip_device_count(void) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    return n;
}
const char* rsrl_hip_last_error(void) { return g_last_error.c_str(); }

int rsrl_hip_config_init(rsrl_hip_config* cfg) {
    if (!cfg) return fail(RSRL_HIP_EINVAL, "null cfg");
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (uint32_t)sizeof(*cfg);
    cfg->domain = RSRL_MOUNTAIN_CAR; cfg->basis = RSRL_FOURIER; cfg->order = 5;
    cfg->n_tilings = 8; cfg->tiles_per_dim = 8;
    cfg->algo = RSRL_QLEARNING; cfg->policy = RSRL_GREEDY;
    cfg->weight_mode = RSRL_W_PER_ENV; cfg->weight_dtype = RSRL_W_F32;
    cfg->n_envs = 1; cfg->seed = 0;
    cfg->gamma = 0.9; cfg->lr = 0.001; cfg->alpha = 1.0; cfg->epsilon = 0.1; cfg->tau = 1.0;
    cfg->max_episode_steps = 0; cfg->steps_per_launch = 0;
    cfg->trace = RSRL_TRACE_ACCUMULATE; cfg->lambda = 0.0; cfg->lr_td = 0.0;
    cfg->agent_policy = -1; cfg->agent_epsilon = 0.1; cfg->agent_tau = 1.0; cfg->exchange = RSRL_EXCHANGE_AUTO;
    cfg->sigma = 0.0; cfg->n_steps = 1;
    cfg->epsilon_decay = 1.0; cfg->epsilon_min = 0.0;
    return RSRL_HIP_OK;
}

int rsrl_hip_destroy(rsrl_hip_ctx* c) {
    if (!c) return RSRL_HIP_OK;
    // (a ctx whose creation failed on its device ordinal is torn down through here too: the failure of this call must not stay behind as the thread's
    //  last HIP error -- tests/fuzz_abi.py found it reported by the next ctx's first launch check)
    if (hipSetDevice(c->cfg.device) != hipSuccess) (void)hipGetLastError();
    if (c->tp.stage) (void)trait_flush(c);       // trait calls accepted but not launched: the caller's arrays are still written
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& ev : c->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& s : c->scratch) if (s.p) (void)hipFree(s.p);
    if (c->state) (void)hipFree(c->state);
    if (c->action) (void)hipFree(c->action);
    if (c->ep_step) (void)hipFree(c->ep_step);
    if (c->W) (void)hipFree(c->W);
    if (c->dW) (void)hipFree(c->dW);
    if (c->dW_rep) (void)hipFree(c->dW_rep);
    if (c->sh_tab) (void)hipFree(c->sh_tab);
    if (c->h_fx) (void)hipFree(c->h_fx);
    if (c->sc_keys) (void)hipFree(c->sc_keys);
    if (c->sc_terms) (void)hipFree(c->sc_terms);
    if (c->W2) (void)hipFree(c->W2);
    if (c->qs_buf) (void)hipFree(c->qs_buf);
    if (c->qs_head) (void)hipFree(c->qs_head);
    if (c->qs_len) (void)hipFree(c->qs_len);
    if (c->step_graph_exec) (void)hipGraphExecDestroy(c->step_graph_exec);
    if (c->step_graph) (void)hipGraphDestroy(c->step_graph);
    if (c->d_t) (void)hipFree(c->d_t);
    if (c->d_dyn) (void)hipFree(c->d_dyn);
    if (c->qcache) (void)hipFree(c->qcache);
    if (c->tq_key) (void)hipFree(c->tq_key);
    if (c->Z) (void)hipFree(c->Z);
    if (c->eps) (void)hipFree(c->eps);
    if (c->flags) (void)hipFree(c->flags);
    if (c->sp_keys) (void)hipFree(c->sp_keys);
    if (c->sp_vals) (void)hipFree(c->sp_vals);
    if (c->sp_len) (void)hipFree(c->sp_len);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    for (size_t r = 0; r < c->peer_ptrs.size(); ++r) if (c->peer_opened[r] && c->peer_ptrs[r]) (void)hipIpcCloseMemHandle(c->peer_ptrs[r]);
    if (c->peer_recv) (void)hipFree(c->peer_recv);
    if (c->d_peer_ptrs) (void)hipFree(c->d_peer_ptrs);
    if (c->d_peer_err) (void)hipFree(c->d_peer_err);
    if (c->px_A) (void)hipFree(c->px_A);
    if (c->px_B && c->px_B_owned) (void)hipFree(c->px_B);
    if (c->d_px_Bptrs) (void)hipFree(c->d_px_Bptrs);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    (void)hipGetLastError();                     // (whatever a release above may have failed with is not the next ctx's business)
    delete c;
    return RSRL_HIP_OK;
}

static int create_impl(const rsrl_hip_config* cfg, rsrl_hip_ctx* c) {
    c->cfg = *cfg;
    switch (cfg->domain) {
    case RSRL_MOUNTAIN_CAR: c->D = 2; c->A = 3; break;
    case RSRL_CART_POLE:    c->D = 4; c->A = 2; break;
    case RSRL_ACROBOT:      c->D = 4; c->A = 3; break;
    default: return fail(RSRL_HIP_EINVAL, "unknown domain %d", cfg->domain);
    }
    if (cfg->n_envs < 1) return fail(RSRL_HIP_EINVAL, "n_envs must be >= 1");
    if (cfg->n_envs + cfg->env_offset > (int64_t)0xffffffffLL || cfg->env_offset < 0)
        return fail(RSRL_HIP_EINVAL, "global env ids must fit 32 bits");
    if (cfg->algo < 0 || cfg->algo > RSRL_Q_SIGMA) return fail(RSRL_HIP_EINVAL, "unknown algo %d", cfg->algo);
    if (cfg->algo == RSRL_Q_SIGMA) {
        // any basis but the order-7 wave family: register-family Fourier, the generic Fourier orders, tile coding (per-learner tables)
        if (cfg->weight_mode != RSRL_W_PER_ENV || cfg->weight_dtype != RSRL_W_F32)
            return fail(RSRL_HIP_EINVAL, "QSigma needs per-learner f32 weights");
        if (!(cfg->sigma >= 0.0 && cfg->sigma <= 1.0)) return fail(RSRL_HIP_EINVAL, "sigma must be in [0, 1]");
        if (cfg->n_steps < 1 || cfg->n_steps > 32) return fail(RSRL_HIP_EINVAL, "n_steps must be in [1, 32]");
    }
    if (cfg->policy < 0 || cfg->policy > RSRL_RANDOM) return fail(RSRL_HIP_EINVAL, "unknown policy %d", cfg->policy);
    // Softmax::new panics for |tau| < 1e-7 (policies/softmax.rs:63-66)
    if (cfg->policy == RSRL_SOFTMAX && std::fabs(cfg->tau) < 1e-7)
        return fail(RSRL_HIP_EINVAL, "Tau parameter in Softmax must be non-zero.");
    if (cfg->weight_dtype != RSRL_W_F32 && cfg->weight_dtype != RSRL_W_BF16) return fail(RSRL_HIP_EINVAL, "unknown weight dtype %d", cfg->weight_dtype);
    if (cfg->agent_policy < -1 || cfg->agent_policy > RSRL_RANDOM) return fail(RSRL_HIP_EINVAL, "unknown agent policy %d", cfg->agent_policy);
    if (cfg->agent_policy == RSRL_SOFTMAX && std::fabs(cfg->agent_tau) < 1e-7)
        return fail(RSRL_HIP_EINVAL, "Tau parameter in Softmax must be non-zero.");
    if (cfg->agent_policy == RSRL_EPSILON_GREEDY && !(cfg->agent_epsilon >= 0.0 && cfg->agent_epsilon <= 1.0))
        return fail(RSRL_HIP_EINVAL, "agent_epsilon must be in [0,1]");
    if (cfg->exchange != RSRL_EXCHANGE_RCCL && cfg->exchange != RSRL_EXCHANGE_PEER && cfg->exchange != RSRL_EXCHANGE_AUTO) return fail(RSRL_HIP_EINVAL, "unknown exchange %d", cfg->exchange);
    if (cfg->basis == RSRL_FOURIER) {
        if (cfg->order < 1 || cfg->order > 7) return fail(RSRL_HIP_EINVAL, "Fourier order must be in [1, 7]");
        c->F = 1; for (int i = 0; i < c->D; ++i) c->F *= (cfg->order + 1);
    } else if (cfg->basis == RSRL_TILE_CODING) {
        if (cfg->tiles_per_dim < 1 || cfg->tiles_per_dim > 64) return fail(RSRL_HIP_EINVAL, "tiles_per_dim must be in [1, 64]");
        int64_t cells = 1; for (int i = 0; i < c->D; ++i) cells *= cfg->tiles_per_dim;
        if (cells * cfg->n_tilings > (int64_t)1 << 30) return fail(RSRL_HIP_EINVAL, "tile table too large");
        // (a shared table is gathered through one 32-bit buffer descriptor)
        if (cfg->weight_mode == RSRL_W_SHARED && cells * cfg->n_tilings * c->A * 4 >= (int64_t)1 << 31) return fail(RSRL_HIP_EINVAL, "a shared tile table must be smaller than 2 GiB");
        c->F = (int)(cells * cfg->n_tilings);
    } else {
        return fail(RSRL_HIP_EINVAL, "unknown basis %d", cfg->basis);
    }
    if (cfg->weight_mode == RSRL_W_SHARED && !is_wave(*cfg) && is_generic_fourier(*cfg))
        return fail(RSRL_HIP_EINVAL, "shared weights need a register-family Fourier order (MountainCar 1-5, CartPole/Acrobot 1) or tile coding");
    if (is_wave(*cfg)) {
        if (cfg->weight_mode == RSRL_W_SHARED) return fail(RSRL_HIP_EINVAL, "shared weights are not available for the order-7 wave family yet");
    } else if (cfg->weight_dtype != RSRL_W_F32) {
        return fail(RSRL_HIP_EINVAL, "bf16 weights are available for Fourier order 7 on CartPole / Acrobot only");
    }
    if (!is_wave(*cfg) && !model_supported(*cfg))
        return fail(RSRL_HIP_EINVAL, "basis %d (order %d / %d tilings) on domain %d has no kernel yet", cfg->basis, cfg->order, cfg->n_tilings, cfg->domain);
    if (is_pred(cfg->algo)) {
        const bool tile_ok = cfg->basis == RSRL_TILE_CODING && cfg->weight_mode == RSRL_W_PER_ENV;
        if (!tile_ok && (cfg->basis != RSRL_FOURIER || cfg->weight_mode != RSRL_W_PER_ENV))
            return fail(RSRL_HIP_EINVAL, "the prediction agents (TD, TDLambda) need per-learner weights on a Fourier basis "
                                         "or on tile coding");
        if (cfg->policy != RSRL_RANDOM) return fail(RSRL_HIP_EINVAL, "prediction agents have no Q function: the behaviour policy must be RSRL_RANDOM");
        if (cfg->algo == RSRL_TD_LAMBDA) {
            if (cfg->trace < 0 || cfg->trace > RSRL_TRACE_DUTCH) return fail(RSRL_HIP_EINVAL, "unknown trace rule %d", cfg->trace);
            if (!(cfg->lambda >= 0.0 && cfg->lambda <= 1.0)) return fail(RSRL_HIP_EINVAL, "lambda must be in [0, 1]");
        }
    }
    if (cfg->algo == RSRL_GREEDY_GQ) {
        if (cfg->weight_mode != RSRL_W_PER_ENV) return fail(RSRL_HIP_EINVAL, "GreedyGQ needs per-learner weights");
        if (!(cfg->lr_td >= 0.0)) return fail(RSRL_HIP_EINVAL, "lr_td must be >= 0");
    }
    if (is_lambda(cfg->algo)) {
        const bool tile_ok = cfg->basis == RSRL_TILE_CODING && cfg->weight_mode == RSRL_W_PER_ENV;     // dense per-learner trace tables
        const bool wave_ok = cfg->basis == RSRL_FOURIER && is_wave(*cfg) && cfg->weight_mode == RSRL_W_PER_ENV;      // (f32, or bf16 + stochastic rounding: round 6)
        // ... or sparse per-learner traces over ONE shared table (traces.rs:5-12 over params/sparse.rs; round 5)
        const bool sparse_ok = is_sparse_lambda(*cfg) && (cfg->n_tilings == 4 || cfg->n_tilings == 8 || cfg->n_tilings == 16);
        if (!tile_ok && !wave_ok && !sparse_ok && (cfg->basis != RSRL_FOURIER || is_wave(*cfg) || cfg->weight_mode != RSRL_W_PER_ENV))
            return fail(RSRL_HIP_EINVAL, "the eligibility-trace agents need per-learner weights on a Fourier basis "
                                         "or on tile coding (per-learner tables, or one shared table with sparse per-learner traces)");
        if (cfg->trace < 0 || cfg->trace > RSRL_TRACE_DUTCH) return fail(RSRL_HIP_EINVAL, "unknown trace rule %d", cfg->trace);
        if (!(cfg->lambda >= 0.0 && cfg->lambda <= 1.0)) return fail(RSRL_HIP_EINVAL, "lambda must be in [0, 1]");
    }
    if (!(cfg->epsilon_decay > 0.0 && cfg->epsilon_decay <= 1.0)) return fail(RSRL_HIP_EINVAL, "epsilon_decay must be in (0, 1] (1 = no schedule)");
    if (!(cfg->epsilon_min >= 0.0 && cfg->epsilon_min <= 1.0)) return fail(RSRL_HIP_EINVAL, "epsilon_min must be in [0, 1]");
    if (cfg->epsilon_decay != 1.0) {
        // the kernels that run the schedule: k_train_reg<.., ESCHED>, k_train_lambda, k_train_mem
        const bool reg = cfg->basis == RSRL_FOURIER && !is_wave(*cfg) && !is_generic_fourier(*cfg);
        const bool one_step = cfg->algo == RSRL_QLEARNING || cfg->algo == RSRL_SARSA || cfg->algo == RSRL_EXPECTED_SARSA || cfg->algo == RSRL_PAL;
        // (round 6: + the order-7 wave family -- k_train_wave / k_train_wave_pk <.., ESCHED>, k_wave_lambda -- f32 and bf16)
        const bool ok = cfg->policy == RSRL_EPSILON_GREEDY && cfg->weight_mode == RSRL_W_PER_ENV && cfg->steps_per_launch != 1 &&
                        (((reg || is_wave(*cfg)) && (one_step || is_lambda(cfg->algo))) || (!reg && !is_wave(*cfg) && one_step));
        if (!ok) return fail(RSRL_HIP_EINVAL, "epsilon_decay (the per-learner epsilon schedule) needs policy = EpsilonGreedy, per-learner weights, steps_per_launch != 1 and "
                                              "a one-step agent or SARSALambda / QLambda on a register-family or order-7 wave-family Fourier basis, or a one-step agent on "
                                              "tile coding / a generic Fourier order");
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(RSRL_HIP_EHIP, "no HIP device");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(RSRL_HIP_EINVAL, "device %d out of range (%d devices)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    { int cus = 0; HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device)); if (cus > 0) { c->n_simd = 4 * cus; c->n_cu = cus; } }
    if (cfg->peer_timeout_ms < 0) return fail(RSRL_HIP_EINVAL, "peer_timeout_ms must be >= 0");
    if (cfg->peer_timeout_ms > 0) c->peer_timeout = (uint64_t)cfg->peer_timeout_ms * 100000ull;
    else if (const char* e = getenv("RSRL_PEER_TIMEOUT_MS")) { const long ms = atol(e); if (ms > 0) c->peer_timeout = (uint64_t)ms * 100000ull; }
    if (cfg->stream) { c->stream = (hipStream_t)cfg->stream; c->own_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    const int64_t N = cfg->n_envs;
    const bool shared = cfg->weight_mode == RSRL_W_SHARED;
    c->w_stride = shared ? 1 : N;
    c->Aw = is_pred(cfg->algo) ? 1 : c->A;
    c->w_elems = (size_t)c->Aw * c->F * (size_t)(shared ? 1 : N);
    // a ctx that steps one batch-step per launch streams W every step: learner-major rows (W[N][A][F]) let k_step_reg_lm
    // write back only the touched column (RSRL_K1_FEATURE_MAJOR=1 keeps the feature-major layout, for A/B runs)
    if (!shared && cfg->steps_per_launch == 1 && cfg->basis == RSRL_FOURIER && !is_wave(*cfg) && !is_generic_fourier(*cfg) &&
        !has_aux(cfg->algo) && !is_pred(cfg->algo) && cfg->algo != RSRL_Q_SIGMA && (c->A * c->F) % 4 == 0 && c->F % 4 == 0 &&
        (uint64_t)c->w_elems * 4ull < (1ull << 32) && !getenv("RSRL_K1_FEATURE_MAJOR")) {
        c->w_stride = 1;
        c->w_ls = (int64_t)c->A * c->F;
        const char* kq = getenv("RSRL_K1_QUAD");
        // four lanes per learner (k_step_reg_q4) pays once there is more than one round of one-lane waves to overlap: measured
        // 19.8 vs 21.3 us per launch at 131 072 learners, 32.7 vs 38.0 at 262 144, but 9.8 vs 9.0 at 65 536 (RSRL_K1_QUAD=1 / 0 forces)
        c->k1_quad = c->A <= 3 && (kq ? kq[0] != '0' : N >= 131072);
    }
    c->dw_elems = (size_t)c->Aw * c->F;
    c->n_stat_slots = is_wave(*cfg) ? wave_grid_for(N) : (c->k1_quad ? (size_t)((N + 63) / 64) : grid_for(N));     // one statistics slot per thread block
    if ((is_lambda(cfg->algo) || is_pred(cfg->algo)) && cfg->basis == RSRL_TILE_CODING) c->n_stat_slots = (size_t)N;      // ... and there a block is a learner
    if (is_lambda(cfg->algo) && is_generic_fourier(*cfg) && !is_wave(*cfg)) c->n_stat_slots = (size_t)((N + 63) / 64);     // k_train_lambda_mem4: 64 learners per block
    HIP_TRY(hipMalloc((void**)&c->state, sizeof(float) * c->D * (size_t)N));
    HIP_TRY(hipMalloc((void**)&c->action, sizeof(int32_t) * (size_t)N));
    HIP_TRY(hipMalloc((void**)&c->ep_step, sizeof(uint32_t) * (size_t)N));
    c->w_bytes = c->w_elems * (cfg->weight_dtype == RSRL_W_BF16 ? 2 : 4);
    HIP_TRY(hipMalloc((void**)&c->W, c->w_bytes));
    HIP_TRY(hipMalloc((void**)&c->dW, sizeof(float) * c->dw_elems));
    HIP_TRY(hipMalloc((void**)&c->qcache, sizeof(float) * c->A * (size_t)N));
    // the trait-granular fast path: learner-major per-learner f32 weights on a basis / agent kernels_trait.hpp is instantiated for, one epsilon for the ctx
    if (c->w_ls != 1 && cfg->weight_dtype == RSRL_W_F32 && cfg->epsilon_decay == 1.0 && trait_lm_available(cfg->domain, cfg->order, cfg->algo) &&
        !getenv("RSRL_NO_TRAIT_FAST"))
        HIP_TRY(hipMalloc((void**)&c->tq_key, sizeof(float) * c->D * (size_t)N));
    if (cfg->algo == RSRL_Q_SIGMA) {
        const size_t nf = (size_t)(c->D + 5) * (size_t)cfg->n_steps * (size_t)N;
        HIP_TRY(hipMalloc((void**)&c->qs_buf, sizeof(float) * nf));
        HIP_TRY(hipMalloc((void**)&c->qs_head, sizeof(uint32_t) * (size_t)N));
        HIP_TRY(hipMalloc((void**)&c->qs_len, sizeof(uint32_t) * (size_t)N));
        HIP_TRY(hipMemsetAsync(c->qs_buf, 0, sizeof(float) * nf, c->stream));
        HIP_TRY(hipMemsetAsync(c->qs_head, 0, sizeof(uint32_t) * (size_t)N, c->stream));
        HIP_TRY(hipMemsetAsync(c->qs_len, 0, sizeof(uint32_t) * (size_t)N, c->stream));          // Backup::new: empty
    }
    if (cfg->epsilon_decay != 1.0) {
        HIP_TRY(hipMalloc((void**)&c->eps, sizeof(float) * (size_t)N));
        hipLaunchKernelGGL(k_fill_f32, dim3(grid_for(N)), dim3(kBlock), 0, c->stream, c->eps, N, (float)cfg->epsilon);
        KCHECK();
    }
    if (is_sparse_lambda(*cfg)) {
        const int64_t slice = (int64_t)(c->F / cfg->n_tilings) * c->A;
        if (slice > 65536) return fail(RSRL_HIP_EINVAL, "SARSALambda / QLambda over a shared tile table: one tiling's slice (cells * actions = %lld entries) must not "
                                                        "exceed 65 536 (16-bit slice-relative keys between the step and the trace kernel)", (long long)slice);
        HIP_TRY(hipMalloc((void**)&c->sp_keys, sizeof(uint32_t) * (size_t)kSparseCap * (size_t)N));
        HIP_TRY(hipMalloc((void**)&c->sp_vals, sizeof(float) * (size_t)kSparseCap * (size_t)N));
        HIP_TRY(hipMalloc((void**)&c->sp_len, sizeof(uint32_t) * (size_t)cfg->n_tilings * (size_t)N));
        HIP_TRY(hipMemsetAsync(c->sp_len, 0, sizeof(uint32_t) * (size_t)cfg->n_tilings * (size_t)N, c->stream));        // Trace::zeros: empty lists
        // (the lists are written only below their lengths; what lies beyond is never read as an entry, but a checkpoint copies whole rows)
        HIP_TRY(hipMemsetAsync(c->sp_keys, 0, sizeof(uint32_t) * (size_t)kSparseCap * (size_t)N, c->stream));
        HIP_TRY(hipMemsetAsync(c->sp_vals, 0, sizeof(float) * (size_t)kSparseCap * (size_t)N, c->stream));
        c->sp_lds = slice * 8 <= 128 * 1024;
        if (c->sp_lds && slice * 8 > 64 * 1024) {                       // more dynamic LDS than a kernel gets by default
            const void* fn = cfg->n_tilings == 4 ? reinterpret_cast<const void*>(&k_sparse_trace_scatter<4>)
                           : cfg->n_tilings == 8 ? reinterpret_cast<const void*>(&k_sparse_trace_scatter<8>) : reinterpret_cast<const void*>(&k_sparse_trace_scatter<16>);
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(slice * 8)) != hipSuccess) { (void)hipGetLastError(); c->sp_lds = false; }
        }
    } else if (has_aux(cfg->algo)) {
        c->z_bytes = c->w_elems * 4;
        HIP_TRY(hipMalloc((void**)&c->Z, c->z_bytes));
        HIP_TRY(hipMemsetAsync(c->Z, 0, c->z_bytes, c->stream));                  // Trace::zeros
    }
    if (shared) {
        HIP_TRY(hipMalloc((void**)&c->flags, (size_t)N));
        if (cfg->basis == RSRL_FOURIER && !is_generic_fourier(*cfg)) {
            c->sh_rows = (unsigned)((N + kSharedBlock - 1) / kSharedBlock);
            HIP_TRY(hipMalloc((void**)&c->sh_tab, sizeof(long long) * 3 * kTabRep * c->dw_elems));
            HIP_TRY(hipMemset(c->sh_tab, 0, sizeof(long long) * 3 * kTabRep * c->dw_elems));
            HIP_TRY(hipMalloc((void**)&c->W2, c->w_bytes));
        }
    }
    HIP_TRY(hipMalloc((void**)&c->d_stats, sizeof(DevStats) * c->n_stat_slots));
    HIP_TRY(hipMalloc((void**)&c->d_t, sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void**)&c->d_dyn, sizeof(DynParams)));
    HIP_TRY(hipHostMalloc((void**)&c->h_stats, sizeof(DevStats) * c->n_stat_slots, hipHostMallocDefault));
    HIP_TRY(hipMemsetAsync(c->W, 0, c->w_bytes, c->stream));                      // LFA::vector zero-initialises
    HIP_TRY(hipMemsetAsync(c->dW, 0, sizeof(float) * c->dw_elems, c->stream));
    // shared tile coding: the mini-batch delta is accumulated in 64-bit fixed point, always.  When one tiling's slice, twice, as
    // 64-bit words fits 128 KiB of LDS the scatter is privatised there and flushed into n_rep copies of the table; otherwise
    // every learner adds its term to ONE copy with device atomics (same integers, same sum).
    if (shared && cfg->basis == RSRL_TILE_CODING) {
        c->tile_slice = (int64_t)(c->F / cfg->n_tilings) * c->A * 16 <= 128 * 1024;
        // copies of the delta table the scatter blocks flush into (RSRL_TILE_REPLICAS tunes it; us per batch-step at 262 144 learners: 1: 25.7,
        // 2: 24.7, 4: 24.1, 8: 24.8, 16: 26.4).  The scatter fused into the step kernel measured 28.7-35.0: scripts/ab/round6_pruned_knobs.patch
        const char* e = getenv("RSRL_TILE_REPLICAS");
        const int r = e ? atoi(e) : 4;
        const bool privatised = c->sp_keys ? c->sp_lds : c->tile_slice;
        c->n_rep = !privatised ? 1 : (r < 1 ? 1 : (r > 16 ? 16 : r));                       // k_apply_rep sums up to 16 copies
        HIP_TRY(hipMalloc((void**)&c->dW_rep, sizeof(long long) * c->dw_elems * c->n_rep));
        HIP_TRY(hipMemsetAsync(c->dW_rep, 0, sizeof(long long) * c->dw_elems * c->n_rep, c->stream));
        if (c->tile_slice || c->sp_keys) {                               // the scatter is a kernel of its own (k_tile_scatter; k_sparse_trace_scatter)
            HIP_TRY(hipMalloc((void**)&c->sc_keys, sizeof(uint16_t) * (size_t)cfg->n_tilings * (size_t)N));
            HIP_TRY(hipMalloc((void**)&c->sc_terms, sizeof(float) * (size_t)N));
        }
    }
    if (shared) {
        HIP_TRY(hipMalloc((void**)&c->d_peer_err, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(c->d_peer_err, 0, sizeof(uint32_t), c->stream));
        HIP_TRY(hipMalloc((void**)&c->h_fx, sizeof(long long) * c->dw_elems));
        HIP_TRY(hipMemsetAsync(c->h_fx, 0, sizeof(long long) * c->dw_elems, c->stream));
    }
    HIP_TRY(hipMemsetAsync(c->action, 0, sizeof(int32_t) * (size_t)N, c->stream));
    HIP_TRY(hipMemsetAsync(c->ep_step, 0, sizeof(uint32_t) * (size_t)N, c->stream));
    return RSRL_HIP_OK;
}

int rsrl_hip_domain_reset(rsrl_hip_ctx* c, const uint8_t* mask);

int rsrl_hip_create(const rsrl_hip_config* cfg, rsrl_hip_ctx** out) {
    if (!cfg || !out) return fail(RSRL_HIP_EINVAL, "null argument");
    // struct_size-versioned: a caller built against an older header passes a shorter struct; the fields it does not know
    // keep the defaults of rsrl_hip_config_init
    if (cfg->struct_size < RSRL_HIP_CONFIG_SIZE_V3 || cfg->struct_size > sizeof(rsrl_hip_config))
        return fail(RSRL_HIP_EINVAL, "config struct_size %u not in [%u, %zu] (ABI mismatch)", cfg->struct_size,
                    RSRL_HIP_CONFIG_SIZE_V3, sizeof(rsrl_hip_config));
    rsrl_hip_config full;
    rsrl_hip_config_init(&full);
    memcpy(&full, cfg, cfg->struct_size);
    full.struct_size = (uint32_t)sizeof(full);
    cfg = &full;
    rsrl_hip_ctx* c = new rsrl_hip_ctx();
    int rc = create_impl(cfg, c);
    if (rc == RSRL_HIP_OK) rc = rsrl_hip_domain_reset(c, nullptr);     // envs start at Domain::default()
    if (rc == RSRL_HIP_OK) { hipError_t e = hipStreamSynchronize(c->stream); if (e != hipSuccess) rc = fail(RSRL_HIP_EHIP, "%s", hipGetErrorString(e)); }
    if (rc != RSRL_HIP_OK) { std::string keep = g_last_error; rsrl_hip_destroy(c); g_last_error = keep; *out = nullptr; return rc; }
    *out = c;
    return RSRL_HIP_OK;
}

int rsrl_hip_sync(rsrl_hip_ctx* c) {
    CHECK_CTX(c); FLUSH(c);
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return peer_check(c);
}

int rsrl_hip_state_dim(const rsrl_hip_ctx* c) { return c ? c->D : RSRL_HIP_EINVAL; }
int rsrl_hip_n_actions(const rsrl_hip_ctx* c) { return c ? c->A : RSRL_HIP_EINVAL; }
int rsrl_hip_n_outputs(const rsrl_hip_ctx* c) { return c ? c->Aw : RSRL_HIP_EINVAL; }
int rsrl_hip_n_features(const rsrl_hip_ctx* c) { return c ? c->F : RSRL_HIP_EINVAL; }
int64_t rsrl_hip_n_envs(const rsrl_hip_ctx* c) { return c ? c->cfg.n_envs : RSRL_HIP_EINVAL; }
uint64_t rsrl_hip_step_count(const rsrl_hip_ctx* c) { return c ? c->t + (uint64_t)c->pending : 0; }
int64_t rsrl_hip_pending_steps(const rsrl_hip_ctx* c) { return c ? c->pending : 0; }

int rsrl_hip_state_bounds(const rsrl_hip_ctx* c, double* lo, double* hi) {
    CHECK_CTX(c);
    if (!lo || !hi) return fail(RSRL_HIP_EINVAL, "null argument");
    for (int i = 0; i < c->D; ++i) {
        switch (c->cfg.domain) {
        case 0: lo[i] = Domain<0>::lo_d(i); hi[i] = Domain<0>::hi_d(i); break;
        case 1: lo[i] = Domain<1>::lo_d(i); hi[i] = Domain<1>::hi_d(i); break;
        default: lo[i] = Domain<2>::lo_d(i); hi[i] = Domain<2>::hi_d(i); break;
        }
    }
    return RSRL_HIP_OK;
}

static void state_limits(const rsrl_hip_ctx* c, float* lo, float* hi) {
    double l[8], h[8];
    (void)rsrl_hip_state_bounds(c, l, h);
    for (int d = 0; d < c->D; ++d) { const double w = 1000.0 * (h[d] - l[d]); lo[d] = (float)(l[d] - w); hi[d] = (float)(h[d] + w); }
}

int rsrl_hip_set_epsilon(rsrl_hip_ctx* c, double eps) {
    CHECK_CTX(c); FLUSH(c);
    if (!(eps >= 0.0 && eps <= 1.0)) return fail(RSRL_HIP_EINVAL, "epsilon must be in [0,1]");   // gen_bool panics otherwise
    c->cfg.epsilon = eps;
    if (c->eps) {                                                       // the field of every learner
        HIP_TRY(hipSetDevice(c->cfg.device));
        hipLaunchKernelGGL(k_fill_f32, dim3(grid_for(c->cfg.n_envs)), dim3(kBlock), 0, c->stream, c->eps, c->cfg.n_envs, (float)eps);
        KCHECK();
    }
    return RSRL_HIP_OK;
}
int rsrl_hip_get_epsilons(rsrl_hip_ctx* c, float* eps_out) {
    CHECK_CTX(c); FLUSH(c); if (!eps_out) return fail(RSRL_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int64_t N = c->cfg.n_envs;
    if (c->eps) {
        HIP_TRY(hipMemcpyAsync(eps_out, c->eps, sizeof(float) * (size_t)N, hipMemcpyDefault, c->stream));
    } else {
        OutBuf<float> ob;
        TRY(stage_out(c, 0, eps_out, (size_t)N, &ob));
        hipLaunchKernelGGL(k_fill_f32, dim3(grid_for(N)), dim3(kBlock), 0, c->stream, ob.dev, N, (float)c->cfg.epsilon);
        KCHECK();
        bool sync = false; TRY(flush_out(c, &ob, &sync));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}

int rsrl_hip_reset(rsrl_hip_ctx* c) {
    CHECK_CTX(c); FLUSH(c);
    c->q_valid = false;
    HIP_TRY(hipSetDevice(c->cfg.device));
    // QSigma: fresh episodes start from an empty n-step backup (as after a terminal transition, q_sigma.rs:154) -- entries of the
    // abandoned trajectories must not be mixed into the first anchor updates of the new ones
    if (c->qs_len) HIP_TRY(hipMemsetAsync(c->qs_len, 0, sizeof(uint32_t) * (size_t)c->cfg.n_envs, c->stream));
    const Common k = make_common(c);
    const BasisGeom g = make_geom(c);
    if (is_pred(c->cfg.algo)) {
        if (!launch_reset_td(c->cfg.domain, dim3(grid_for(k.n_envs)), dim3(kBlock), c->stream, k, c->t)) return NO_MODEL(c);
    } else if (is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using T = decltype(tag); using WT = typename T::wt;
            hipLaunchKernelGGL((k_wave_reset<T::domain, WT>), dim3(wave_grid_for(k.n_envs)), dim3(kBlock), 0, c->stream, k, (const WT*)c->W, c->t);
        });
    } else if (!for_model(c, [&](auto tag) {
            using M = typename decltype(tag)::type;
            hipLaunchKernelGGL((k_reset<M>), dim3(grid_for(k.n_envs)), dim3(kBlock), 0, c->stream, k, g, c->t);
        })) return NO_MODEL(c);
    KCHECK();
    return RSRL_HIP_OK;
}

int rsrl_hip_get_states(rsrl_hip_ctx* c, float* states) {
    CHECK_CTX(c); FLUSH(c); if (!states) return fail(RSRL_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(states, c->state, sizeof(float) * c->D * (size_t)c->cfg.n_envs, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return peer_check(c);
}
int rsrl_hip_set_states(rsrl_hip_ctx* c, const float* states) {
    CHECK_CTX(c); FLUSH(c);
    if (!states) return fail(RSRL_HIP_EINVAL, "null argument");
    TRY(check_host_states(c, states, (size_t)c->cfg.n_envs));
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (is_device_ptr(states)) {
        // same rule as for a host array, checked where the data is; a refused array leaves the ctx untouched
        StateLimits lim; state_limits(c, lim.lo, lim.hi);
        TRY(scratch_reserve(c, 7, sizeof(unsigned)));
        unsigned* d_bad = (unsigned*)c->scratch[7].p;
        unsigned bad = 0;
        HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(k_check_states, dim3(grid_for(c->cfg.n_envs)), dim3(kBlock), 0, c->stream, states, c->cfg.n_envs, c->D, lim, d_bad);
        KCHECK();
        HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (bad) return fail(RSRL_HIP_EINVAL, "%u component(s) of the device array of states are not finite values within 1000 widths of their dimension's bounds", bad);
    }
    c->q_valid = false;
    HIP_TRY(hipMemcpyAsync(c->state, states, sizeof(float) * c->D * (size_t)c->cfg.n_envs, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}
int rsrl_hip_get_actions(rsrl_hip_ctx* c, int32_t* actions) {
    CHECK_CTX(c); FLUSH(c); if (!actions) return fail(RSRL_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(actions, c->action, sizeof(int32_t) * (size_t)c->cfg.n_envs, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}
int rsrl_hip_set_actions(rsrl_hip_ctx* c, const int32_t* actions) {
    CHECK_CTX(c); FLUSH(c); if (!actions) return fail(RSRL_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    TRY(check_host_actions(actions, (size_t)c->cfg.n_envs, c->A));
    HIP_TRY(hipMemcpyAsync(c->action, actions, sizeof(int32_t) * (size_t)c->cfg.n_envs, hipMemcpyDefault, c->stream));
    hipLaunchKernelGGL(k_clamp_actions, dim3(grid_for(c->cfg.n_envs)), dim3(kBlock), 0, c->stream, c->action, c->cfg.n_envs, c->A);
    KCHECK();
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}

// ---- ABI 8: the learners' state between two driver calls that is neither weights nor env state (include/rsrl_hip.h)
static bool carries_q(const rsrl_hip_ctx* c) {          // the kernels that read Common::qcache: the register family's one-step loops
    const int al = c->cfg.algo;
    return c->cfg.basis == RSRL_FOURIER && !is_wave(c->cfg) && !is_generic_fourier(c->cfg) && c->cfg.weight_mode == RSRL_W_PER_ENV &&
           (al == RSRL_QLEARNING || al == RSRL_SARSA || al == RSRL_EXPECTED_SARSA || al == RSRL_PAL);
}
int rsrl_hip_get_episode_steps(rsrl_hip_ctx* c, uint32_t* steps) {
    CHECK_CTX(c); FLUSH(c); if (!steps) return fail(RSRL_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(steps, c->ep_step, sizeof(uint32_t) * (size_t)c->cfg.n_envs, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return peer_check(c);
}
int rsrl_hip_set_episode_steps(rsrl_hip_ctx* c, const uint32_t* steps) {
    CHECK_CTX(c); FLUSH(c); if (!steps) return fail(RSRL_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(c->ep_step, steps, sizeof(uint32_t) * (size_t)c->cfg.n_envs, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}
int rsrl_hip_get_q_carry(rsrl_hip_ctx* c, float* q, int32_t* valid) {
    CHECK_CTX(c); FLUSH(c); if (!q || !valid) return fail(RSRL_HIP_EINVAL, "null argument");
    *valid = (carries_q(c) && c->q_valid) ? 1 : 0;
    if (!*valid) return RSRL_HIP_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(q, c->qcache, sizeof(float) * (size_t)c->A * (size_t)c->cfg.n_envs, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}
int rsrl_hip_set_q_carry(rsrl_hip_ctx* c, const float* q) {
    CHECK_CTX(c); FLUSH(c); if (!q) return fail(RSRL_HIP_EINVAL, "null argument");
    if (!carries_q(c)) return fail(RSRL_HIP_EINVAL, "this ctx's kernels evaluate Q(s,.) from the weights every step: there is nothing carried to restore");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(c->qcache, q, sizeof(float) * (size_t)c->A * (size_t)c->cfg.n_envs, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->q_valid = true;
    return RSRL_HIP_OK;
}

int rsrl_hip_domain_step(rsrl_hip_ctx* c, const int32_t* actions, float* from_states, float* next_states,
                         float* rewards, uint8_t* terminal) {
    CHECK_CTX(c); FLUSH(c);
    c->q_valid = false;
    HIP_TRY(hipSetDevice(c->cfg.device));
    // the trait-granular loop on a ctx-owned stream: a transition handed over in device arrays is ACCEPTED here and launched with the calls that
    // follow it (rsrl_hip_handle on exactly these arrays, rsrl_hip_domain_reset on the terminal flags, rsrl_hip_policy_sample of the ctx's envs) as
    // one kernel -- or, by whatever other call comes next, as the kernel it would have been now.  Same results, same order (kernels_trait.hpp).
    if (trait_fast(c) && c->own_stream && actions && from_states && next_states && rewards && terminal && is_device_ptr(actions) &&
        is_device_ptr(from_states) && is_device_ptr(next_states) && is_device_ptr(rewards) && is_device_ptr(terminal) && !getenv("RSRL_NO_TRAIT_DEFER")) {
        c->tp = rsrl_hip_ctx::TraitPend{};
        c->tp.stage = 1; c->tp.act = actions; c->tp.from = from_states; c->tp.to = next_states; c->tp.rew = rewards; c->tp.term = terminal;
        return RSRL_HIP_OK;
    }
    const int64_t N = c->cfg.n_envs; const size_t DN = (size_t)c->D * N;
    const int32_t* d_act; OutBuf<float> ofrom, onext, orew; OutBuf<uint8_t> oterm;
    TRY(check_host_actions(actions, (size_t)N, c->A));
    TRY(stage_in(c, 0, actions, (size_t)N, &d_act));
    TRY(stage_out(c, 1, from_states, DN, &ofrom));
    TRY(stage_out(c, 2, next_states, DN, &onext));
    TRY(stage_out(c, 3, rewards, (size_t)N, &orew));
    TRY(stage_out(c, 4, terminal, (size_t)N, &oterm));
    const Common k = make_common(c);
    TRY(launch_domain_step(c, k, d_act, ofrom.dev, onext.dev, orew.dev, oterm.dev));
    bool sync = false;
    TRY(flush_out(c, &ofrom, &sync)); TRY(flush_out(c, &onext, &sync));
    TRY(flush_out(c, &orew, &sync)); TRY(flush_out(c, &oterm, &sync));
    if (sync || (actions && !is_device_ptr(actions))) HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}

int rsrl_hip_domain_reset(rsrl_hip_ctx* c, const uint8_t* mask) {
    CHECK_CTX(c);
    if (c->tp.stage == 2 && mask && mask == c->tp.term) { c->tp.stage = 3; return RSRL_HIP_OK; }      // the new episodes of the transition just handled
    FLUSH(c);
    c->q_valid = false;
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int64_t N = c->cfg.n_envs;
    const uint8_t* d_mask;
    TRY(stage_in(c, 0, mask, (size_t)N, &d_mask));
    const Common k = make_common(c);
    TRY(launch_domain_reset(c, k, d_mask));
    if (mask && !is_device_ptr(mask)) HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}

static int qop(rsrl_hip_ctx* c, int op, const float* states, int64_t M_, float* fout, size_t fcount, int32_t* iout,
               size_t icount = 0, const float* fin = nullptr, size_t fin_count = 0, const int32_t* iin = nullptr, uint64_t step_t = 0) {
    CHECK_CTX(c); FLUSH(c);
    if (!states || M_ < 1 || M_ > c->cfg.n_envs) return fail(RSRL_HIP_EINVAL, "bad batch (M=%lld, n_envs=%lld)", (long long)M_, (long long)c->cfg.n_envs);
    if (is_pred(c->cfg.algo) && op != QOP_EVALUATE && op != QOP_FEATURES)
        return fail(RSRL_HIP_ESTATE, "a prediction agent has a state-value function only (use rsrl_hip_q_evaluate for V(s))");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const float* d_states; OutBuf<float> of; OutBuf<int32_t> oi;
    TRY(stage_in(c, 0, states, (size_t)c->D * M_, &d_states));
    TRY(stage_out(c, 1, fout, fcount, &of));
    TRY(stage_out(c, 2, iout, icount ? icount : (size_t)M_, &oi));
    const float* d_fin = nullptr; const int32_t* d_iin = nullptr;
    if (iin) TRY(check_host_actions(iin, (size_t)M_, c->A));
    TRY(stage_in(c, 3, fin, fin_count, &d_fin));
    TRY(stage_in(c, 4, iin, (size_t)M_, &d_iin));
    const Common k = make_common(c);
    const uint64_t call = (op == QOP_SAMPLE_STEP || op == QOP_SAMPLE_INIT) ? step_t : c->api_calls;      // (the driver loop's sample: addressed by the batch-step)
    if (op == QOP_SAMPLE) c->api_calls++;
    const BasisGeom g = make_geom(c);
    if (is_pred(c->cfg.algo) && op == QOP_EVALUATE && is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using T = decltype(tag); using WT = typename T::wt;
            hipLaunchKernelGGL((k_wave_v_evaluate<T::domain, WT>), dim3(wave_grid_for(M_)), dim3(kBlock), 0, c->stream, (const WT*)c->W, d_states, M_, of.dev);
        });
    } else if (is_pred(c->cfg.algo) && op == QOP_EVALUATE && c->cfg.basis == RSRL_TILE_CODING) {
        if (!launch_td_tile(c->cfg.domain, c->cfg.n_tilings, false, 0, c->stream, k, g, make_td(c), 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, M_,
                            of.dev, d_states)) return NO_MODEL(c);
    } else if (is_pred(c->cfg.algo) && op == QOP_EVALUATE && is_generic_fourier(c->cfg)) {
        if (!launch_td_model(c->cfg, dim3(grid_for(M_)), dim3(kBlock), c->stream, k, make_td(c), g, false, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, M_,
                             of.dev, d_states)) return NO_MODEL(c);
    } else if (is_pred(c->cfg.algo) && op == QOP_EVALUATE) {
        if (!launch_v_evaluate(c->cfg.domain, c->cfg.order, dim3(grid_for(M_)), dim3(kBlock), c->stream, k, d_states, M_, of.dev)) return NO_MODEL(c);
    } else if (is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using T = decltype(tag); using WT = typename T::wt;
            hipLaunchKernelGGL((k_wave_qop<T::domain, WT>), dim3(wave_grid_for(M_)), dim3(kBlock), 0, c->stream, k, (const WT*)c->W, op, d_states, M_, call, of.dev, oi.dev,
                               d_fin, d_iin);
        });
    } else if (!for_model(c, [&](auto tag) {
            using M = typename decltype(tag)::type;
            hipLaunchKernelGGL((k_qop<M>), dim3(grid_for(M_)), dim3(kBlock), 0, c->stream, k, g, op, d_states, M_, call, of.dev, oi.dev, d_fin, d_iin);
        })) return NO_MODEL(c);
    KCHECK();
    bool sync = !is_device_ptr(states) || (fin && !is_device_ptr(fin)) || (iin && !is_device_ptr(iin));
    TRY(flush_out(c, &of, &sync)); TRY(flush_out(c, &oi, &sync));
    if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}

int rsrl_hip_q_find_min(rsrl_hip_ctx* c, const float* states, int64_t M, int32_t* idx_out, float* val_out) {
    return qop(c, QOP_FIND_MIN, states, M, val_out, (size_t)M, idx_out);
}
int rsrl_hip_q_expected_value(rsrl_hip_ctx* c, const float* states, int64_t M, const float* probs, float* out) {
    if (!probs || !out) return fail(RSRL_HIP_EINVAL, "null argument");
    return qop(c, QOP_EXPECTED, states, M, out, (size_t)M, nullptr, 0, probs, c ? (size_t)c->A * M : 0);
}
int rsrl_hip_policy_prob(rsrl_hip_ctx* c, const float* states, const int32_t* actions, int64_t M, float* prob_out) {
    if (!actions || !prob_out) return fail(RSRL_HIP_EINVAL, "null argument");
    return qop(c, QOP_PROB_SA, states, M, prob_out, (size_t)M, nullptr, 0, nullptr, 0, actions);
}
int rsrl_hip_q_evaluate(rsrl_hip_ctx* c, const float* states, int64_t M, float* q_out) {
    if (!q_out) return fail(RSRL_HIP_EINVAL, "null argument");
    return qop(c, QOP_EVALUATE, states, M, q_out, c ? (size_t)c->Aw * M : 0, nullptr);
}
int rsrl_hip_q_find_max(rsrl_hip_ctx* c, const float* states, int64_t M, int32_t* idx_out, float* val_out) {
    return qop(c, QOP_FIND_MAX, states, M, val_out, (size_t)M, idx_out);
}
// states == NULL: policy.sample(rng, env.emit().state()) for the ctx's OWN envs (M = n_envs) -- the driver loop's behaviour sample: it draws what
// batch-step step_count - 1 of rsrl_hip_train draws (the initial sample's stream before the first handle), and the actions also become the ctx's pending ones
static int sample_emit(rsrl_hip_ctx* c, int64_t M, int32_t* actions_out) {
    if (M != c->cfg.n_envs) return fail(RSRL_HIP_EINVAL, "policy_sample(states = NULL) samples for the ctx's own envs: M must be n_envs (%lld), got %lld", (long long)c->cfg.n_envs, (long long)M);
    if (is_pred(c->cfg.algo)) return fail(RSRL_HIP_ESTATE, "a prediction agent has a state-value function only (use rsrl_hip_q_evaluate for V(s))");
    const bool dev_out = is_device_ptr(actions_out);
    if (c->tp.stage == 3 && dev_out) {
        // the whole batch-step -- transition, handle, new episodes, sample -- as ONE kernel
        const rsrl_hip_ctx::TraitPend p = c->tp;
        c->tp.stage = 0;
        HIP_TRY(hipSetDevice(c->cfg.device));
        TRY(trait_cache_ready(c));
        const Common k = make_common(c);
        TraitIo io{};
        io.act = p.act; io.td_out = p.td; io.o_from = p.from; io.o_to = p.to; io.o_rew = p.rew; io.o_term = p.term; io.o_act = actions_out;
        io.qkey = c->tq_key; io.Mn = M;
        TRY(timing_begin(c));
        if (!launch_trait_lm(c->cfg.domain, c->cfg.order, c->cfg.algo, c->cfg.policy, c->stream, k, io, p.t_handle)) return NO_MODEL(c);
        KCHECK();
        c->kernel_name = "k_trait_lm<step>";
        return timing_end(c);
    }
    FLUSH(c);
    HIP_TRY(hipSetDevice(c->cfg.device));
    const uint64_t t = c->t ? c->t - 1 : 0;
    const uint32_t blk = c->t ? BLK_STEP : BLK_INIT;
    if (trait_fast(c)) {
        OutBuf<int32_t> oa;
        TRY(stage_out(c, 2, actions_out, (size_t)M, &oa));
        TRY(trait_cache_ready(c));
        const Common k = make_common(c);
        TRY(timing_begin(c));
        if (!launch_trait_sample(c->cfg.domain, c->cfg.order, c->stream, k, nullptr, M, t, blk, c->tq_key, oa.dev)) return NO_MODEL(c);
        KCHECK();
        TRY(timing_end(c));
        bool sync = false;
        TRY(flush_out(c, &oa, &sync));
        if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
        return RSRL_HIP_OK;
    }
    TRY(qop(c, c->t ? QOP_SAMPLE_STEP : QOP_SAMPLE_INIT, c->state, M, nullptr, 0, actions_out, 0, nullptr, 0, nullptr, t));
    HIP_TRY(hipMemcpyAsync(c->action, actions_out, sizeof(int32_t) * (size_t)M, hipMemcpyDefault, c->stream));
    if (!dev_out) HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}
int rsrl_hip_policy_sample(rsrl_hip_ctx* c, const float* states, int64_t M, int32_t* actions_out) {
    CHECK_CTX(c);
    if (!actions_out) return fail(RSRL_HIP_EINVAL, "null argument");
    if (!states) return sample_emit(c, M, actions_out);
    if (trait_fast(c) && M >= 1 && M <= c->cfg.n_envs) {
        // the fast path's sample: a state the hand-over cache holds costs 20 B instead of the learner's 432 B of weights; same bits either way
        FLUSH(c);
        HIP_TRY(hipSetDevice(c->cfg.device));
        const float* d_states; OutBuf<int32_t> oa;
        TRY(stage_in(c, 0, states, (size_t)c->D * M, &d_states));
        TRY(stage_out(c, 2, actions_out, (size_t)M, &oa));
        TRY(trait_cache_ready(c));
        const Common k = make_common(c);
        const uint64_t call = c->api_calls++;
        if (!launch_trait_sample(c->cfg.domain, c->cfg.order, c->stream, k, d_states, M, call, BLK_API, c->tq_key, oa.dev)) return NO_MODEL(c);
        KCHECK();
        bool sync = !is_device_ptr(states);
        TRY(flush_out(c, &oa, &sync));
        if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
        return RSRL_HIP_OK;
    }
    return qop(c, QOP_SAMPLE, states, M, nullptr, 0, actions_out);
}
int rsrl_hip_policy_mode(rsrl_hip_ctx* c, const float* states, int64_t M, int32_t* actions_out) {
    CHECK_CTX(c);
    if (!actions_out) return fail(RSRL_HIP_EINVAL, "null argument");
    if (c->cfg.policy == RSRL_RANDOM) return fail(RSRL_HIP_EINVAL, "Random policy has no mode.");   // random.rs:47
    return qop(c, QOP_MODE, states, M, nullptr, 0, actions_out);
}
int rsrl_hip_policy_probs(rsrl_hip_ctx* c, const float* states, int64_t M, float* probs_out) {
    if (!probs_out) return fail(RSRL_HIP_EINVAL, "null argument");
    return qop(c, QOP_PROBS, states, M, probs_out, c ? (size_t)c->A * M : 0, nullptr);
}

int rsrl_hip_project(rsrl_hip_ctx* c, const float* states, int64_t M, float* phi_out) {
    CHECK_CTX(c);
    if (!phi_out) return fail(RSRL_HIP_EINVAL, "null argument");
    if (c->cfg.basis != RSRL_FOURIER) return fail(RSRL_HIP_EINVAL, "dense projection needs a Fourier basis");
    return qop(c, QOP_FEATURES, states, M, phi_out, (size_t)c->F * M, nullptr);
}

int rsrl_hip_tile_indices(rsrl_hip_ctx* c, const float* states, int64_t M, int32_t* idx_out) {
    CHECK_CTX(c);
    if (!idx_out) return fail(RSRL_HIP_EINVAL, "null argument");
    if (c->cfg.basis != RSRL_TILE_CODING) return fail(RSRL_HIP_EINVAL, "tile indices need a tile-coding basis");
    return qop(c, QOP_FEATURES, states, M, nullptr, 0, idx_out, (size_t)c->cfg.n_tilings * M);
}

int rsrl_hip_handle(rsrl_hip_ctx* c, const float* from_states, const int32_t* actions, const float* rewards,
                    const float* to_states, const uint8_t* terminal, int64_t M, float* td_error_out) {
    CHECK_CTX(c);
    if (!from_states || !actions || !rewards || !to_states || !terminal) return fail(RSRL_HIP_EINVAL, "null argument");
    if (M < 1 || M > c->cfg.n_envs) return fail(RSRL_HIP_EINVAL, "bad batch size");
    // the transition rsrl_hip_domain_step has just been handed (same arrays, every learner): accepted, launched with it (rsrl_hip_domain_step)
    if (c->tp.stage == 1 && from_states == c->tp.from && actions == c->tp.act && rewards == c->tp.rew && to_states == c->tp.to && terminal == c->tp.term &&
        M == c->cfg.n_envs && (!td_error_out || is_device_ptr(td_error_out))) {
        c->tp.stage = 2; c->tp.td = td_error_out; c->tp.t_handle = c->t;
        c->t += 1;
        return RSRL_HIP_OK;
    }
    FLUSH(c);
    if (c->st_rccl_group) return fail(RSRL_HIP_ESTATE, "this ctx is a rank of a single-thread RCCL group: handle() all-reduces the mini-batch delta, and one thread "
                                                       "cannot issue that for one rank at a time -- use one thread / process per rank, or RSRL_EXCHANGE_PEER");
    HIP_TRY(hipSetDevice(c->cfg.device));
    TRY(check_host_actions(actions, (size_t)M, c->A));
    const float *d_from, *d_rew, *d_to; const int32_t* d_act; const uint8_t* d_term; OutBuf<float> otd;
    const bool all_device = is_device_ptr(from_states) && is_device_ptr(actions) && is_device_ptr(rewards) && is_device_ptr(to_states) && is_device_ptr(terminal) &&
                            (!td_error_out || is_device_ptr(td_error_out));
    TRY(stage_in(c, 0, from_states, (size_t)c->D * M, &d_from));
    TRY(stage_in(c, 1, actions, (size_t)M, &d_act));
    TRY(stage_in(c, 2, rewards, (size_t)M, &d_rew));
    TRY(stage_in(c, 3, to_states, (size_t)c->D * M, &d_to));
    TRY(stage_in(c, 4, terminal, (size_t)M, &d_term));
    TRY(stage_out(c, 5, td_error_out, (size_t)M, &otd));
    const Common k = make_common(c);
    const BasisGeom g = make_geom(c);
    if (is_sparse_lambda(c->cfg)) {
        // transition i is LEARNER i's (round 6): its residual against the shared table, its trace, the mini-batch's delta -- the driver loop's three launches on
        // the caller's transitions (kernels_sparse_lambda.hpp)
        const float step_size = (float)c->cfg.alpha;
        Common ks = k;
        ks.alg.kind = c->cfg.algo == RSRL_SARSA_LAMBDA ? ALG_SARSA : ALG_QLEARNING; ks.alg.lr = step_size;
        const int slice = (int)((int64_t)(c->F / c->cfg.n_tilings) * c->A);
        if (!for_model(c, [&](auto tag) {
                using Mo = typename decltype(tag)::type;
                if constexpr (Mo::kSparse) {
                    hipLaunchKernelGGL((k_sparse_handle<Mo>), dim3(grid_for(M)), dim3(kBlock), 0, c->stream, ks, g, d_from, d_act, d_rew, d_to, d_term, M, c->t,
                                       c->cfg.algo == RSRL_Q_LAMBDA ? 1 : 0, c->flags, c->sc_keys, c->sc_terms, otd.dev);
                    const int per = 512;
                    hipLaunchKernelGGL((k_sparse_trace_scatter<Mo::kT>), dim3((unsigned)((M + per - 1) / per), (unsigned)Mo::kT), dim3(1024), c->sp_lds ? (size_t)slice * 8 : 0,
                                       c->stream, c->sc_keys, c->sc_terms, c->flags, SparseTrace{c->sp_keys, c->sp_vals, c->sp_len}, make_lambda(c), M,
                                       (int64_t)c->cfg.n_envs, slice, per, c->dW_rep, c->n_rep, (int64_t)c->dw_elems, FxScale(step_size).inv_lsb, c->sp_lds ? 1 : 0);
                }
            })) return NO_MODEL(c);
        KCHECK();
        const int n = (int)c->dw_elems;
        hipLaunchKernelGGL(k_apply_rep, dim3(((n + 1) / 2 + 255) / 256), dim3(256), 0, c->stream, c->multi ? (float*)nullptr : c->W, c->dW, c->dW_rep, c->n_rep, n,
                           tile_lsb(step_size));
        KCHECK();
        if (c->multi) {
            TRY(exchange_dw(c, c->t, nullptr, k.xdelta));
            if (c->cfg.exchange == RSRL_EXCHANGE_PEER) c->peer_seq += 1;
            hipLaunchKernelGGL(k_apply_dw, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->W, c->dW, n);
            KCHECK();
        }
        c->q_valid = false; c->tq_valid = false;
        c->t += 1;
        bool sync = !all_device;
        TRY(flush_out(c, &otd, &sync));
        if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
        return RSRL_HIP_OK;
    }
    if (trait_fast(c)) {
        TRY(launch_trait_handle(c, k, d_from, d_act, d_rew, d_to, d_term, M, c->t, otd.dev));
    } else if (is_wave(c->cfg) && is_wave_aux_algo(c->cfg.algo)) {
        launch_wave_aux(c, dim3(wave_grid_for(M)), k, make_wave_aux(c), c->t, 1, (DevStats*)nullptr, d_from, d_act, d_rew, d_to, d_term, M, otd.dev);
    } else if (is_pred(c->cfg.algo) && c->cfg.basis == RSRL_TILE_CODING) {
        if (!launch_td_tile(c->cfg.domain, c->cfg.n_tilings, c->cfg.algo == RSRL_TD_LAMBDA, M, c->stream, k, g, make_td(c), c->t, 1, nullptr, d_from, d_rew,
                            d_to, d_term, M, otd.dev, nullptr)) return NO_MODEL(c);
    } else if (is_pred(c->cfg.algo) && is_generic_fourier(c->cfg)) {
        if (!launch_td_model(c->cfg, dim3(grid_for(M)), dim3(kBlock), c->stream, k, make_td(c), g, c->cfg.algo == RSRL_TD_LAMBDA, c->t, 1, nullptr, d_from, d_rew,
                             d_to, d_term, M, otd.dev, nullptr)) return NO_MODEL(c);
    } else if (is_pred(c->cfg.algo)) {
        if (!launch_handle_td(c->cfg.domain, c->cfg.order, c->cfg.algo == RSRL_TD_LAMBDA, dim3(grid_for(M)), dim3(kBlock), c->stream, k, make_td(c),
                              d_from, d_rew, d_to, d_term, M, otd.dev)) return NO_MODEL(c);
    } else if (c->cfg.algo == RSRL_Q_SIGMA && is_wave(c->cfg)) {
        if (c->cfg.domain == RSRL_CART_POLE) hipLaunchKernelGGL((k_wave_qsigma<1>), dim3(wave_grid_for(M)), dim3(kBlock), 0, c->stream, k, make_qs(c), c->t, 1, (DevStats*)nullptr, d_from, d_act, d_rew, d_to, d_term, M, otd.dev);
        else hipLaunchKernelGGL((k_wave_qsigma<2>), dim3(wave_grid_for(M)), dim3(kBlock), 0, c->stream, k, make_qs(c), c->t, 1, (DevStats*)nullptr, d_from, d_act, d_rew, d_to, d_term, M, otd.dev);
    } else if (c->cfg.algo == RSRL_Q_SIGMA) {
        const bool reg = c->cfg.basis == RSRL_FOURIER && !is_generic_fourier(c->cfg);
        if (!(reg ? launch_qsigma(c->cfg.domain, c->cfg.order, dim3(grid_for(M)), dim3(kBlock), c->stream, k, make_qs(c), g, c->t, 0, nullptr,
                                  d_from, d_act, d_rew, d_to, d_term, M, otd.dev)
                  : launch_qsigma_model(c->cfg, dim3(grid_for(M)), dim3(kBlock), c->stream, k, make_qs(c), g, c->t, 0, nullptr,
                                        d_from, d_act, d_rew, d_to, d_term, M, otd.dev))) return NO_MODEL(c);
    } else if (c->cfg.algo == RSRL_GREEDY_GQ) {
        const bool reg = c->cfg.basis == RSRL_FOURIER && !is_generic_fourier(c->cfg);
        if (!(reg ? launch_handle_gq(c->cfg.domain, c->cfg.order, dim3(grid_for(M)), dim3(kBlock), c->stream, k, make_gq(c),
                                     d_from, d_act, d_rew, d_to, d_term, M, otd.dev)
                  : launch_gq_model(c->cfg, dim3(grid_for(M)), dim3(kBlock), c->stream, k, make_gq(c), g, c->t, 0, nullptr,
                                    d_from, d_act, d_rew, d_to, d_term, M, otd.dev))) return NO_MODEL(c);
    } else if (is_lambda(c->cfg.algo) && c->cfg.basis == RSRL_TILE_CODING) {
        if (!launch_lambda_tile(c->cfg.domain, c->cfg.n_tilings, M, c->stream, k, g, make_lambda(c), c->t, 1, nullptr, d_from, d_act, d_rew, d_to, d_term,
                                M, otd.dev)) return NO_MODEL(c);
    } else if (is_lambda(c->cfg.algo) && is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using T = decltype(tag); using WT = typename T::wt;
            hipLaunchKernelGGL((k_wave_lambda<T::domain, WT>), dim3(wave_grid_for(M)), dim3(kBlock), 0, c->stream, k, make_lambda(c), (WT*)c->W, c->t, 1, (DevStats*)nullptr,
                               d_from, d_act, d_rew, d_to, d_term, M, otd.dev);
        });
    } else if (is_lambda(c->cfg.algo) && is_generic_fourier(c->cfg)) {
        if (!launch_lambda_model(c->cfg, dim3(grid_for(M)), dim3(kBlock), c->stream, k, make_lambda(c), g, c->t, 1, nullptr, d_from, d_act, d_rew, d_to, d_term,
                                 M, otd.dev)) return NO_MODEL(c);
    } else if (is_lambda(c->cfg.algo)) {
        if (!launch_handle_lambda(c->cfg.domain, c->cfg.order, dim3(grid_for(M)), dim3(kBlock), c->stream, k, make_lambda(c),
                                  d_from, d_act, d_rew, d_to, d_term, M, c->t, otd.dev)) return NO_MODEL(c);
    } else if (is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using T = decltype(tag); using WT = typename T::wt;
            hipLaunchKernelGGL((k_wave_handle<T::domain, WT>), dim3(wave_grid_for(M)), dim3(kBlock), 0, c->stream, k, (WT*)c->W, d_from, d_act, d_rew, d_to, d_term, M, c->t, otd.dev);
        });
    } else if (!for_model(c, [&](auto tag) {
            using Mo = typename decltype(tag)::type;
            hipLaunchKernelGGL((k_handle<Mo>), dim3(grid_for(M)), dim3(kBlock), 0, c->stream, k, g, d_from, d_act, d_rew, d_to, d_term, M, c->t, otd.dev, c->h_fx);
        })) return NO_MODEL(c);
    KCHECK();
    if (c->cfg.weight_mode == RSRL_W_SHARED) {
        // the mini-batch delta (accumulated in fixed point: exact, reproducible) of ALL ranks is applied by every rank (replicas
        // of W stay bit-identical): same exchange step as inside rsrl_hip_train
        const int n = (int)c->dw_elems;
        hipLaunchKernelGGL(k_fx_finalize, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->h_fx, c->dW, n, tile_lsb((float)c->cfg.lr));
        KCHECK();
        TRY(exchange_dw(c, c->t, nullptr, k.xdelta));
        if (c->multi && c->cfg.exchange == RSRL_EXCHANGE_PEER) c->peer_seq += 1;
        hipLaunchKernelGGL(k_apply_dw, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->W, c->dW, n);
        KCHECK();
    }
    c->q_valid = false;
    if (!trait_fast(c)) c->tq_valid = false;
    c->t += 1;          // one handle call = one batch-step of learning: the agent-side draws (SARSA's inner sample,
                        // bf16 stochastic rounding) advance exactly as they do inside rsrl_hip_train
    bool sync = !all_device;   // host inputs are staged asynchronously: they must have been read when the call returns; device arrays are asynchronous
    TRY(flush_out(c, &otd, &sync));
    if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}

int rsrl_hip_get_weights(rsrl_hip_ctx* c, int64_t env_index, float* w) {
    CHECK_CTX(c); FLUSH(c); if (!w) return fail(RSRL_HIP_EINVAL, "null argument");
    const bool shared = c->cfg.weight_mode == RSRL_W_SHARED;
    if (!shared && (env_index < 0 || env_index >= c->cfg.n_envs)) return fail(RSRL_HIP_EINVAL, "env_index out of range");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int n = c->F * c->Aw; OutBuf<float> ow;
    TRY(stage_out(c, 0, w, (size_t)n, &ow));
    if (is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using WT = typename decltype(tag)::wt;
            hipLaunchKernelGGL((k_wave_weights_get<WT>), dim3((n + 255) / 256), dim3(256), 0, c->stream, (const WT*)c->W + env_index * (int64_t)n, c->F, c->Aw, ow.dev);
        });
    } else
    hipLaunchKernelGGL(k_weights_get, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->W, c->cfg.basis == RSRL_TILE_CODING, c->w_stride, (shared ? 0 : env_index) * c->w_ls, c->F, c->Aw, ow.dev);
    KCHECK();
    bool sync = false; TRY(flush_out(c, &ow, &sync));
    if (sync) { HIP_TRY(hipStreamSynchronize(c->stream)); return peer_check(c); }      // a failed exchange must not pass for weights
    return RSRL_HIP_OK;
}
int rsrl_hip_set_weights(rsrl_hip_ctx* c, int64_t env_index, const float* w) {
    CHECK_CTX(c); FLUSH(c);
    c->q_valid = false; c->tq_valid = false; if (!w) return fail(RSRL_HIP_EINVAL, "null argument");
    const bool shared = c->cfg.weight_mode == RSRL_W_SHARED;
    if (!shared && (env_index < 0 || env_index >= c->cfg.n_envs)) return fail(RSRL_HIP_EINVAL, "env_index out of range");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int n = c->F * c->Aw; const float* d_w;
    TRY(stage_in(c, 0, w, (size_t)n, &d_w));
    if (is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using WT = typename decltype(tag)::wt;
            const int64_t groups = (int64_t)c->Aw * (c->F / 8);
            hipLaunchKernelGGL((k_wave_weights_set<WT>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, c->stream, (WT*)c->W, env_index, (int64_t)1, c->F, c->Aw, d_w);
        });
    } else
    hipLaunchKernelGGL(k_weights_set, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->W, c->cfg.basis == RSRL_TILE_CODING, c->w_stride, (shared ? 0 : env_index) * c->w_ls, c->F, c->Aw, d_w);
    KCHECK();
    if (!is_device_ptr(w)) HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}
static int traces_rw(rsrl_hip_ctx* c, int64_t env_index, float* out, const float* in) {
    CHECK_CTX(c); FLUSH(c);
    if (c->sp_keys) {
        // a learner's SPARSE trace over the shared table, shown as the dense (F, A) matrix it stands for; the list itself is not settable
        if (!out) return fail(RSRL_HIP_ESTATE, "the sparse traces of a shared-table lambda agent cannot be set from a dense matrix");
        if (env_index < 0 || env_index >= c->cfg.n_envs) return fail(RSRL_HIP_EINVAL, "env_index out of range");
        HIP_TRY(hipSetDevice(c->cfg.device));
        const int n = c->F * c->Aw;
        OutBuf<float> oz;
        TRY(stage_out(c, 0, out, (size_t)n, &oz));
        HIP_TRY(hipMemsetAsync(oz.dev, 0, sizeof(float) * (size_t)n, c->stream));
        hipLaunchKernelGGL(k_sparse_trace_get, dim3(kSparseCap / 256), dim3(256), 0, c->stream, SparseTrace{c->sp_keys, c->sp_vals, c->sp_len}, c->cfg.n_tilings, env_index, oz.dev);
        KCHECK();
        bool sync = false; TRY(flush_out(c, &oz, &sync));
        if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
        return RSRL_HIP_OK;
    }
    if (!c->Z) return fail(RSRL_HIP_ESTATE, "this agent has no auxiliary matrix (eligibility trace / fa_td weights)");
    if (env_index < 0 || env_index >= c->cfg.n_envs) return fail(RSRL_HIP_EINVAL, "env_index out of range");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int n = c->F * c->Aw;
    if (out) {
        OutBuf<float> oz;
        TRY(stage_out(c, 0, out, (size_t)n, &oz));
        if (is_wave(c->cfg)) hipLaunchKernelGGL((k_wave_weights_get<float>), dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float*)c->Z + env_index * (int64_t)n, c->F, c->Aw, oz.dev);
        else hipLaunchKernelGGL(k_weights_get, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->Z, c->cfg.basis == RSRL_TILE_CODING, c->w_stride, env_index, c->F, c->Aw, oz.dev);
        KCHECK();
        bool sync = false; TRY(flush_out(c, &oz, &sync));
        if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
        const float* d_z;
        TRY(stage_in(c, 0, in, (size_t)n, &d_z));
        if (is_wave(c->cfg)) hipLaunchKernelGGL((k_wave_weights_set<float>), dim3((unsigned)(((int64_t)c->Aw * (c->F / 8) + 255) / 256)), dim3(256), 0, c->stream, c->Z, env_index, (int64_t)1, c->F, c->Aw, d_z);
        else hipLaunchKernelGGL(k_weights_set, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->Z, c->cfg.basis == RSRL_TILE_CODING, c->w_stride, env_index, c->F, c->Aw, d_z);
        KCHECK();
        if (!is_device_ptr(in)) HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return RSRL_HIP_OK;
}
int rsrl_hip_get_traces(rsrl_hip_ctx* c, int64_t env_index, float* z) {
    if (!z) return fail(RSRL_HIP_EINVAL, "null argument");
    CHECK_CTX(c);
    if (!is_lambda(c->cfg.algo) && c->cfg.algo != RSRL_TD_LAMBDA) return fail(RSRL_HIP_ESTATE, "this agent has no eligibility trace");
    return traces_rw(c, env_index, z, nullptr);
}
int rsrl_hip_set_traces(rsrl_hip_ctx* c, int64_t env_index, const float* z) {
    if (!z) return fail(RSRL_HIP_EINVAL, "null argument");
    CHECK_CTX(c);
    if (!is_lambda(c->cfg.algo) && c->cfg.algo != RSRL_TD_LAMBDA) return fail(RSRL_HIP_ESTATE, "this agent has no eligibility trace");
    return traces_rw(c, env_index, nullptr, z);
}
int rsrl_hip_get_td_weights(rsrl_hip_ctx* c, int64_t env_index, float* v) {
    if (!v) return fail(RSRL_HIP_EINVAL, "null argument");
    CHECK_CTX(c);
    if (c->cfg.algo != RSRL_GREEDY_GQ) return fail(RSRL_HIP_ESTATE, "only GreedyGQ has a second approximator (fa_td)");
    return traces_rw(c, env_index, v, nullptr);
}
int rsrl_hip_set_td_weights(rsrl_hip_ctx* c, int64_t env_index, const float* v) {
    if (!v) return fail(RSRL_HIP_EINVAL, "null argument");
    CHECK_CTX(c);
    if (c->cfg.algo != RSRL_GREEDY_GQ) return fail(RSRL_HIP_ESTATE, "only GreedyGQ has a second approximator (fa_td)");
    return traces_rw(c, env_index, nullptr, v);
}

// ---- checkpoint: header + every learner's weights in the reference (F, A) order -----------------------------------
// The header is serialised FIELD BY FIELD (little-endian, no implicit padding); layout in include/rsrl_hip.h.
namespace {
constexpr uint32_t kCkptVersion = 3;          // files carrying aux_kind 3 (QSigma's n-step backups); every other file is still written as version 2
constexpr uint32_t kCkptVersionEps = 4;       // ... or as version 4 when the ctx runs the per-learner epsilon schedule: f32 eps[N] follows the payload
constexpr uint32_t kCkptVersionSparse = 6;    // files carrying aux_kind 4 (the sparse per-learner traces over a shared table): u64 n_envs, u64 env_offset, u32 len[N], lists
constexpr uint32_t kCkptVersionSparse5 = 5;   // ... as round 5 wrote them (no n_envs / env_offset in front of the lengths): still read
constexpr int64_t kSparseChunk = 4096;        // learners per staging chunk of the sparse lists
constexpr size_t kCkptHeaderBytes = 72;
struct Ckpt {
    int32_t domain, basis, order, n_tilings, tiles_per_dim, weight_mode, F, A, algo, weight_dtype, aux_kind;
    int64_t n_learners; uint64_t step_count;
    bool has_eps;                                 // (not a header field: the file version says it)
};
// 1 = eligibility traces, 2 = fa_td weights (both: a second matrix of W's shape), 3 = QSigma's per-learner n-step backups,
// 4 = every learner's sparse trace over the shared table (the lists, compact)
int aux_kind_of(const rsrl_hip_ctx* c) { return c->sp_keys ? 4 : (c->qs_buf ? 3 : (!c->Z ? 0 : (c->cfg.algo == RSRL_GREEDY_GQ ? 2 : 1))); }
size_t qs_floats(const rsrl_hip_ctx* c) { return (size_t)(c->D + 5) * (size_t)c->cfg.n_steps * (size_t)c->cfg.n_envs; }
Ckpt ckpt_of(const rsrl_hip_ctx* c) {
    Ckpt h{};
    h.domain = c->cfg.domain; h.basis = c->cfg.basis; h.order = c->cfg.order; h.n_tilings = c->cfg.n_tilings;
    h.tiles_per_dim = c->cfg.tiles_per_dim; h.weight_mode = c->cfg.weight_mode; h.F = c->F; h.A = c->Aw;
    h.algo = c->cfg.algo; h.weight_dtype = c->cfg.weight_dtype; h.aux_kind = aux_kind_of(c);
    h.n_learners = c->cfg.weight_mode == RSRL_W_SHARED ? 1 : c->cfg.n_envs; h.step_count = c->t;
    h.has_eps = c->eps != nullptr;
    return h;
}
void put32(uint8_t*& p, uint32_t v) { for (int i = 0; i < 4; ++i) *p++ = (uint8_t)(v >> (8 * i)); }
void put64(uint8_t*& p, uint64_t v) { for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(v >> (8 * i)); }
uint32_t get32(const uint8_t*& p) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)*p++ << (8 * i); return v; }
uint64_t get64(const uint8_t*& p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)*p++ << (8 * i); return v; }
void ckpt_encode(const Ckpt& h, uint8_t (&buf)[kCkptHeaderBytes]) {
    uint8_t* p = buf;
    memcpy(p, "RSRLHIPW", 8); p += 8;
    put32(p, h.has_eps ? kCkptVersionEps : (h.aux_kind == 4 ? kCkptVersionSparse : (h.aux_kind == 3 ? kCkptVersion : 2u)));
    const int32_t f[11] = {h.domain, h.basis, h.order, h.n_tilings, h.tiles_per_dim, h.weight_mode, h.F, h.A, h.algo, h.weight_dtype, h.aux_kind};
    for (int32_t v : f) put32(p, (uint32_t)v);
    put64(p, (uint64_t)h.n_learners); put64(p, h.step_count);
}
bool ckpt_decode(const uint8_t (&buf)[kCkptHeaderBytes], Ckpt* h, uint32_t* version) {
    const uint8_t* p = buf;
    if (memcmp(p, "RSRLHIPW", 8) != 0) return false;
    p += 8;
    *version = get32(p);
    int32_t* f[11] = {&h->domain, &h->basis, &h->order, &h->n_tilings, &h->tiles_per_dim, &h->weight_mode, &h->F, &h->A, &h->algo, &h->weight_dtype, &h->aux_kind};
    for (int32_t* v : f) *v = (int32_t)get32(p);
    h->n_learners = (int64_t)get64(p); h->step_count = get64(p);
    h->has_eps = *version == kCkptVersionEps;
    return true;
}
}  // namespace
static int traces_rw(rsrl_hip_ctx* c, int64_t env_index, float* out, const float* in);
int rsrl_hip_save_weights(rsrl_hip_ctx* c, const char* path) {
    CHECK_CTX(c); FLUSH(c);
    if (!path) return fail(RSRL_HIP_EINVAL, "null path");
    FILE* f = fopen(path, "wb");
    if (!f) return fail(RSRL_HIP_EINVAL, "cannot open %s for writing", path);
    const Ckpt h = ckpt_of(c);
    uint8_t hdr[kCkptHeaderBytes]; ckpt_encode(h, hdr);
    int rc = RSRL_HIP_OK;
    if (fwrite(hdr, 1, sizeof(hdr), f) != sizeof(hdr)) rc = fail(RSRL_HIP_EINVAL, "short write to %s", path);
    std::vector<float> w((size_t)c->F * c->Aw);
    for (int pass = 0; pass < ((h.aux_kind == 1 || h.aux_kind == 2) ? 2 : 1); ++pass)            // every learner's weights, then every learner's auxiliary matrix
        for (int64_t i = 0; rc == RSRL_HIP_OK && i < h.n_learners; ++i) {
            rc = pass == 0 ? rsrl_hip_get_weights(c, i, w.data()) : traces_rw(c, i, w.data(), nullptr);
            if (rc == RSRL_HIP_OK && fwrite(w.data(), sizeof(float), w.size(), f) != w.size()) rc = fail(RSRL_HIP_EINVAL, "short write to %s", path);
        }
    if (rc == RSRL_HIP_OK && h.aux_kind == 3) {                        // QSigma: ring heads, lengths, entries (SoA [field][slot][learner])
        const size_t N = (size_t)c->cfg.n_envs, nf = qs_floats(c);
        std::vector<uint32_t> hl(2 * N); std::vector<float> buf(nf);
        hipError_t e = hipMemcpyAsync(hl.data(), c->qs_head, 4 * N, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hl.data() + N, c->qs_len, 4 * N, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(buf.data(), c->qs_buf, 4 * nf, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(RSRL_HIP_EHIP, "reading the QSigma backups: %s", hipGetErrorString(e));
        else if (fwrite(hl.data(), 4, 2 * N, f) != 2 * N || fwrite(buf.data(), 4, nf, f) != nf) rc = fail(RSRL_HIP_EINVAL, "short write to %s", path);
    }
    if (rc == RSRL_HIP_OK && h.aux_kind == 4) {
        // sparse traces: u64 n_envs, u64 env_offset (whose learners these are), u32 len[N], then per learner its len keys and its len values -- the
        // sub-lists concatenated in tiling order (a key says which tiling it belongs to: the file does not depend on the cap per tiling)
        const int64_t N = c->cfg.n_envs; const int T = c->cfg.n_tilings, cap = kSparseCap / T;
        std::vector<uint32_t> lens((size_t)N * T), tot((size_t)N), keys((size_t)(kSparseChunk * kSparseCap));
        std::vector<float> vals((size_t)(kSparseChunk * kSparseCap));
        hipError_t e = hipMemcpyAsync(lens.data(), c->sp_len, 4 * (size_t)N * T, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(RSRL_HIP_EHIP, "reading the sparse traces: %s", hipGetErrorString(e));
        for (int64_t i = 0; rc == RSRL_HIP_OK && i < N; ++i) {
            uint32_t sum = 0;
            for (int t = 0; t < T; ++t) {
                if (lens[(size_t)i * T + t] > (uint32_t)cap) rc = fail(RSRL_HIP_ESTATE, "learner %lld's sparse trace has %u entries in tiling %d", (long long)i, lens[(size_t)i * T + t], t);
                sum += lens[(size_t)i * T + t];
            }
            tot[(size_t)i] = sum;
        }
        uint8_t who[16]; uint8_t* wp = who; put64(wp, (uint64_t)N); put64(wp, (uint64_t)c->cfg.env_offset);
        if (rc == RSRL_HIP_OK && (fwrite(who, 1, 16, f) != 16 || fwrite(tot.data(), 4, (size_t)N, f) != (size_t)N)) rc = fail(RSRL_HIP_EINVAL, "short write to %s", path);
        std::vector<uint32_t> kk((size_t)kSparseCap); std::vector<float> vv((size_t)kSparseCap);
        for (int64_t i0 = 0; rc == RSRL_HIP_OK && i0 < N; i0 += kSparseChunk) {
            const int64_t n = std::min<int64_t>(kSparseChunk, N - i0);
            e = hipMemcpyAsync(keys.data(), c->sp_keys + i0 * kSparseCap, 4 * (size_t)(n * kSparseCap), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(vals.data(), c->sp_vals + i0 * kSparseCap, 4 * (size_t)(n * kSparseCap), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { rc = fail(RSRL_HIP_EHIP, "reading the sparse traces: %s", hipGetErrorString(e)); break; }
            for (int64_t i = 0; rc == RSRL_HIP_OK && i < n; ++i) {
                size_t l = 0;
                for (int t = 0; t < T; ++t)
                    for (uint32_t j = 0; j < lens[(size_t)(i0 + i) * T + t]; ++j, ++l) {
                        kk[l] = keys[(size_t)(i * kSparseCap + t * cap) + j]; vv[l] = vals[(size_t)(i * kSparseCap + t * cap) + j];
                    }
                if (fwrite(kk.data(), 4, l, f) != l || fwrite(vv.data(), 4, l, f) != l) rc = fail(RSRL_HIP_EINVAL, "short write to %s", path);
            }
        }
    }
    if (rc == RSRL_HIP_OK && h.has_eps) {                              // the schedule's state: every learner's current epsilon
        std::vector<float> e((size_t)c->cfg.n_envs);
        rc = rsrl_hip_get_epsilons(c, e.data());
        if (rc == RSRL_HIP_OK && fwrite(e.data(), 4, e.size(), f) != e.size()) rc = fail(RSRL_HIP_EINVAL, "short write to %s", path);
    }
    if (fclose(f) != 0 && rc == RSRL_HIP_OK) rc = fail(RSRL_HIP_EINVAL, "closing %s failed", path);
    return rc;
}
int rsrl_hip_load_weights(rsrl_hip_ctx* c, const char* path) {
    CHECK_CTX(c); FLUSH(c);
    if (!path) return fail(RSRL_HIP_EINVAL, "null path");
    HIP_TRY(hipSetDevice(c->cfg.device));
    FILE* f = fopen(path, "rb");
    if (!f) return fail(RSRL_HIP_EINVAL, "cannot open %s", path);
    const Ckpt want = ckpt_of(c);
    Ckpt h{}; uint32_t version = 0; uint8_t hdr[kCkptHeaderBytes];
    int rc = RSRL_HIP_OK;
    if (fread(hdr, 1, sizeof(hdr), f) != sizeof(hdr) || !ckpt_decode(hdr, &h, &version)) rc = fail(RSRL_HIP_EINVAL, "%s is not a rsrl_hip weight file", path);
    else if (version != kCkptVersion && version != 2u && version != kCkptVersionEps && version != kCkptVersionSparse && version != kCkptVersionSparse5)
        rc = fail(RSRL_HIP_EINVAL, "%s has checkpoint version %u, this library reads versions 2, %u, %u, %u and %u", path, version, kCkptVersion, kCkptVersionEps,
                  kCkptVersionSparse5, kCkptVersionSparse);
    // a QSigma file written before the backups travelled (version 2, aux_kind 0) is still read: the weights are loaded and the run
    // resumes from EMPTY n-step backups, as after a terminal transition (q_sigma.rs:154)
    // (the same for a sparse-trace file of ABI 7's first build, version 2 / aux_kind 0: the run resumes from EMPTY lists, Trace::zeros)
    const bool old_qsigma = rc == RSRL_HIP_OK && (want.aux_kind == 3 || want.aux_kind == 4) && h.aux_kind == 0 && version == 2u;
    if (rc == RSRL_HIP_OK &&
        (h.domain != want.domain || h.basis != want.basis || h.order != want.order || h.n_tilings != want.n_tilings ||
         h.tiles_per_dim != want.tiles_per_dim || h.weight_mode != want.weight_mode || h.F != want.F || h.A != want.A ||
         h.algo != want.algo || h.weight_dtype != want.weight_dtype || (h.aux_kind != want.aux_kind && !old_qsigma) || h.n_learners != want.n_learners ||
         h.has_eps != want.has_eps))
        rc = fail(RSRL_HIP_EINVAL, "%s was written by a different configuration%s", path,
                  h.has_eps != want.has_eps ? " (the per-learner epsilon schedule, config.epsilon_decay, is part of it)" : "");
    const size_t per = (size_t)c->F * c->Aw;
    std::vector<uint32_t> sp_len_in, sp_len_t;      // sparse traces: a learner's entries in the file; its sub-lists' lengths on the device
    long sp_prefix = 0;
    if (rc == RSRL_HIP_OK) {                                             // a truncated file is refused before anything is touched
        long long expect = (long long)kCkptHeaderBytes + (long long)((h.aux_kind == 1 || h.aux_kind == 2) ? 2 : 1) * h.n_learners * (long long)per * 4 +
                           (h.aux_kind == 3 ? (long long)c->cfg.n_envs * 8 + (long long)qs_floats(c) * 4 : 0) +
                           (h.has_eps ? (long long)c->cfg.n_envs * 4 : 0);
        if (h.aux_kind == 4) {                                           // the lists are compact: their lengths say how long the file is
            const size_t N = (size_t)c->cfg.n_envs;
            sp_len_in.resize(N);
            sp_prefix = version == kCkptVersionSparse ? 16 : 0;
            uint8_t who[16];
            if (fseek(f, (long)(kCkptHeaderBytes + h.n_learners * (long long)per * 4), SEEK_SET) != 0 || (sp_prefix && fread(who, 1, 16, f) != 16))
                rc = fail(RSRL_HIP_EINVAL, "%s is truncated (the sparse traces' owner)", path);
            if (rc == RSRL_HIP_OK && sp_prefix) {                            // whose lists these are: the writer's shard, not only its size
                const uint8_t* wp = who; const uint64_t n_in = get64(wp), off_in = get64(wp);
                if (n_in != (uint64_t)N || off_in != (uint64_t)c->cfg.env_offset)
                    rc = fail(RSRL_HIP_EINVAL, "%s was written by a different configuration (sparse traces of %llu learners at env_offset %llu; this ctx: %zu at %lld)", path,
                              (unsigned long long)n_in, (unsigned long long)off_in, N, (long long)c->cfg.env_offset);
            }
            if (rc == RSRL_HIP_OK && fread(sp_len_in.data(), 4, N, f) != N) rc = fail(RSRL_HIP_EINVAL, "%s is truncated (the sparse traces' lengths)", path);
            expect += sp_prefix + 4 * (long long)N;
            for (size_t i = 0; rc == RSRL_HIP_OK && i < N; ++i) {
                if (sp_len_in[i] > (uint32_t)kSparseCap) rc = fail(RSRL_HIP_EINVAL, "%s: corrupt sparse trace of learner %zu (%u entries)", path, i, sp_len_in[i]);
                expect += 8 * (long long)sp_len_in[i];
            }
        }
        if (rc == RSRL_HIP_OK && (fseek(f, 0, SEEK_END) != 0 || ftell(f) != expect || fseek(f, (long)kCkptHeaderBytes, SEEK_SET) != 0))
            rc = fail(RSRL_HIP_EINVAL, "%s is truncated or has trailing bytes (expected %lld bytes)", path, expect);
    }
    if (rc != RSRL_HIP_OK) { fclose(f); return rc; }
    // staged: the file goes into shadow copies of W (and of the auxiliary matrix); the ctx switches to them only when
    // every learner has been read -- a failing load leaves the ctx exactly as it was
    float* W_old = c->W; float* Z_old = c->Z; float* W_new = nullptr; float* Z_new = nullptr;
    hipError_t e = hipMalloc((void**)&W_new, c->w_bytes);
    if (e == hipSuccess && Z_old) e = hipMalloc((void**)&Z_new, c->z_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(W_new, W_old, c->w_bytes, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess && Z_old) e = hipMemcpyAsync(Z_new, Z_old, c->z_bytes, hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) {
        if (W_new) (void)hipFree(W_new);
        if (Z_new) (void)hipFree(Z_new);
        fclose(f);
        return fail(e == hipErrorOutOfMemory ? RSRL_HIP_ENOMEM : RSRL_HIP_EHIP, "staging buffers for %s: %s", path, hipGetErrorString(e));
    }
    c->W = W_new; c->Z = Z_new;
    std::vector<float> w(per);
    for (int pass = 0; pass < ((h.aux_kind == 1 || h.aux_kind == 2) ? 2 : 1); ++pass)
        for (int64_t i = 0; rc == RSRL_HIP_OK && i < h.n_learners; ++i) {
            if (fread(w.data(), sizeof(float), per, f) != per) { rc = fail(RSRL_HIP_EINVAL, "%s: read error", path); break; }
            rc = pass == 0 ? rsrl_hip_set_weights(c, i, w.data()) : traces_rw(c, i, nullptr, w.data());
        }
    uint32_t* spk_new = nullptr; float* spv_new = nullptr;              // sparse traces: shadow lists, switched in at the end like W
    if (rc == RSRL_HIP_OK && h.aux_kind == 4) {
        const int64_t N = c->cfg.n_envs;
        hipError_t e2 = hipMalloc((void**)&spk_new, 4 * (size_t)kSparseCap * (size_t)N);
        if (e2 == hipSuccess) e2 = hipMalloc((void**)&spv_new, 4 * (size_t)kSparseCap * (size_t)N);
        if (e2 != hipSuccess) rc = fail(e2 == hipErrorOutOfMemory ? RSRL_HIP_ENOMEM : RSRL_HIP_EHIP, "staging buffers for the sparse traces: %s", hipGetErrorString(e2));
        std::vector<uint32_t> keys((size_t)(kSparseChunk * kSparseCap)), kk((size_t)kSparseCap);
        std::vector<float> vals((size_t)(kSparseChunk * kSparseCap)), vv((size_t)kSparseCap);
        if (rc == RSRL_HIP_OK && fseek(f, sp_prefix + 4 * (long)N, SEEK_CUR) != 0) rc = fail(RSRL_HIP_EINVAL, "%s: read error", path);      // (owner and lengths: read above)
        const int T = c->cfg.n_tilings, cap = kSparseCap / T;
        const uint32_t n_keys = (uint32_t)c->F * (uint32_t)c->Aw, slice = n_keys / (uint32_t)T;
        sp_len_t.assign((size_t)N * T, 0u);
        for (int64_t i0 = 0; rc == RSRL_HIP_OK && i0 < N; i0 += kSparseChunk) {
            const int64_t n = std::min<int64_t>(kSparseChunk, N - i0);
            std::fill(keys.begin(), keys.end(), 0u); std::fill(vals.begin(), vals.end(), 0.0f);
            for (int64_t i = 0; rc == RSRL_HIP_OK && i < n; ++i) {
                const size_t l = sp_len_in[(size_t)(i0 + i)];
                if (fread(kk.data(), 4, l, f) != l || fread(vv.data(), 4, l, f) != l) rc = fail(RSRL_HIP_EINVAL, "%s: read error", path);
                for (size_t k = 0; rc == RSRL_HIP_OK && k < l; ++k) {         // every entry into the sub-list of its key's tiling
                    if (kk[k] >= n_keys) { rc = fail(RSRL_HIP_EINVAL, "%s: corrupt sparse trace of learner %lld (key out of range)", path, (long long)(i0 + i)); break; }
                    const uint32_t t = kk[k] / slice; uint32_t& lt = sp_len_t[(size_t)(i0 + i) * T + t];
                    if (lt >= (uint32_t)cap) { rc = fail(RSRL_HIP_EINVAL, "%s: learner %lld's sparse trace holds more than %d entries of tiling %u (this library keeps "
                                                                            "%d entries per learner as %d per tiling)", path, (long long)(i0 + i), cap, t, kSparseCap, cap); break; }
                    keys[(size_t)(i * kSparseCap + (int64_t)t * cap) + lt] = kk[k]; vals[(size_t)(i * kSparseCap + (int64_t)t * cap) + lt] = vv[k];
                    lt += 1;
                }
            }
            if (rc != RSRL_HIP_OK) break;
            e2 = hipMemcpyAsync(spk_new + i0 * kSparseCap, keys.data(), 4 * (size_t)(n * kSparseCap), hipMemcpyHostToDevice, c->stream);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(spv_new + i0 * kSparseCap, vals.data(), 4 * (size_t)(n * kSparseCap), hipMemcpyHostToDevice, c->stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);                 // (the staging vectors are reused by the next chunk)
            if (e2 != hipSuccess) rc = fail(RSRL_HIP_EHIP, "installing the sparse traces: %s", hipGetErrorString(e2));
        }
    }
    std::vector<uint32_t> hl; std::vector<float> ring;
    if (rc == RSRL_HIP_OK && h.aux_kind == 3) {                        // read first, install only when everything has been read
        const size_t N = (size_t)c->cfg.n_envs, nf = qs_floats(c);
        hl.resize(2 * N); ring.resize(nf);
        if (fread(hl.data(), 4, 2 * N, f) != 2 * N || fread(ring.data(), 4, nf, f) != nf) rc = fail(RSRL_HIP_EINVAL, "%s: read error", path);
        for (size_t i = 0; rc == RSRL_HIP_OK && i < N; ++i)
            if (hl[i] >= (uint32_t)c->cfg.n_steps || hl[N + i] > (uint32_t)c->cfg.n_steps) rc = fail(RSRL_HIP_EINVAL, "%s: corrupt QSigma backup of learner %zu", path, i);
    }
    std::vector<float> eps_in;
    if (rc == RSRL_HIP_OK && h.has_eps) {
        eps_in.resize((size_t)c->cfg.n_envs);
        if (fread(eps_in.data(), 4, eps_in.size(), f) != eps_in.size()) rc = fail(RSRL_HIP_EINVAL, "%s: read error", path);
        for (size_t i = 0; rc == RSRL_HIP_OK && i < eps_in.size(); ++i)
            if (!(eps_in[i] >= 0.0f && eps_in[i] <= 1.0f)) rc = fail(RSRL_HIP_EINVAL, "%s: epsilon of learner %zu is outside [0, 1]", path, i);
    }
    fclose(f);
    (void)hipStreamSynchronize(c->stream);
    if (rc == RSRL_HIP_OK && old_qsigma && c->sp_len) {                // old file: no lists in it -> empty ones
        hipError_t e2 = hipMemsetAsync(c->sp_len, 0, sizeof(uint32_t) * (size_t)c->cfg.n_tilings * (size_t)c->cfg.n_envs, c->stream);
        if (e2 != hipSuccess) rc = fail(RSRL_HIP_EHIP, "clearing the sparse traces: %s", hipGetErrorString(e2));
    } else if (rc == RSRL_HIP_OK && old_qsigma) {                      // old file: no backups in it -> empty ones
        hipError_t e2 = hipMemsetAsync(c->qs_len, 0, sizeof(uint32_t) * (size_t)c->cfg.n_envs, c->stream);
        if (e2 == hipSuccess) e2 = hipMemsetAsync(c->qs_head, 0, sizeof(uint32_t) * (size_t)c->cfg.n_envs, c->stream);
        if (e2 != hipSuccess) rc = fail(RSRL_HIP_EHIP, "clearing the QSigma backups: %s", hipGetErrorString(e2));
    }
    if (rc == RSRL_HIP_OK && h.has_eps) {
        hipError_t e2 = hipMemcpyAsync(c->eps, eps_in.data(), 4 * eps_in.size(), hipMemcpyHostToDevice, c->stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
        if (e2 != hipSuccess) rc = fail(RSRL_HIP_EHIP, "installing the learners' epsilons: %s", hipGetErrorString(e2));
    }
    if (rc == RSRL_HIP_OK && h.aux_kind == 3) {
        const size_t N = (size_t)c->cfg.n_envs;
        hipError_t e2 = hipMemcpyAsync(c->qs_head, hl.data(), 4 * N, hipMemcpyHostToDevice, c->stream);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(c->qs_len, hl.data() + N, 4 * N, hipMemcpyHostToDevice, c->stream);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(c->qs_buf, ring.data(), 4 * ring.size(), hipMemcpyHostToDevice, c->stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
        if (e2 != hipSuccess) rc = fail(RSRL_HIP_EHIP, "installing the QSigma backups: %s", hipGetErrorString(e2));
    }
    if (rc == RSRL_HIP_OK && h.aux_kind == 4) {                        // the last step that can fail: the lengths
        hipError_t e2 = hipMemcpyAsync(c->sp_len, sp_len_t.data(), 4 * sp_len_t.size(), hipMemcpyHostToDevice, c->stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
        if (e2 != hipSuccess) rc = fail(RSRL_HIP_EHIP, "installing the sparse traces: %s", hipGetErrorString(e2));
    }
    if (rc == RSRL_HIP_OK) {
        (void)hipFree(W_old); if (Z_old) (void)hipFree(Z_old);
        if (spk_new) { (void)hipFree(c->sp_keys); (void)hipFree(c->sp_vals); c->sp_keys = spk_new; c->sp_vals = spv_new; }
        c->t = h.step_count; c->q_valid = false; c->tq_valid = false;
    } else {
        std::string keep = g_last_error;
        c->W = W_old; c->Z = Z_old;
        (void)hipFree(W_new); if (Z_new) (void)hipFree(Z_new);
        if (spk_new) (void)hipFree(spk_new);
        if (spv_new) (void)hipFree(spv_new);
        g_last_error = keep;
    }
    return rc;
}

int rsrl_hip_set_weights_all(rsrl_hip_ctx* c, const float* w) {
    CHECK_CTX(c); FLUSH(c);
    c->q_valid = false; c->tq_valid = false; if (!w) return fail(RSRL_HIP_EINVAL, "null argument");
    if (c->cfg.weight_mode == RSRL_W_SHARED) return rsrl_hip_set_weights(c, 0, w);
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int n = c->F * c->Aw; const float* d_w;
    TRY(stage_in(c, 0, w, (size_t)n, &d_w));
    const int gy = n < 1024 ? n : 1024;
    if (is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using WT = typename decltype(tag)::wt;
            const int64_t groups = c->cfg.n_envs * (int64_t)c->Aw * (c->F / 8);
            hipLaunchKernelGGL((k_wave_weights_set<WT>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, c->stream, (WT*)c->W, (int64_t)0, c->cfg.n_envs, c->F, c->Aw, d_w);
        });
    } else
    hipLaunchKernelGGL(k_weights_set_all, dim3(grid_for(c->cfg.n_envs), gy), dim3(kBlock), 0, c->stream, c->W, c->cfg.basis == RSRL_TILE_CODING, c->cfg.n_envs, c->cfg.basis == RSRL_TILE_CODING ? c->cfg.n_envs : c->w_stride, c->w_ls,
                       c->F, c->Aw, d_w);
    KCHECK();
    if (!is_device_ptr(w)) HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}

// ---- the fused driver loop -----------------------------------------------------------------
static int timing_begin(rsrl_hip_ctx* c) {
    if (!c->timing) return RSRL_HIP_OK;
    if (c->events_used == c->events.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
        c->events.emplace_back(a, b);
    }
    HIP_TRY(hipEventRecord(c->events[c->events_used].first, c->stream));
    return RSRL_HIP_OK;
}
static int timing_end(rsrl_hip_ctx* c, uint32_t launches) {
    if (!c->timing) return RSRL_HIP_OK;
    HIP_TRY(hipEventRecord(c->events[c->events_used].second, c->stream));
    if (c->event_launches.size() <= c->events_used) c->event_launches.resize(c->events_used + 1);
    c->event_launches[c->events_used] = launches;
    c->events_used++;
    return RSRL_HIP_OK;
}

// shared weights: one batch-step = [phase C of the previous step + phase A] in one launch ->
// delta finalize (+ apply when there is a single rank) -> [all-reduce over ranks -> apply]; the last step of a
// train call is closed by a stand-alone phase C (enqueue_shared_c).
// t_dev != nullptr: the launch is a graph node, t is its offset to the device-side batch-step counter.
// what the step kernel's prologue folds into the weights: 1 = this rank's own delta table (single rank), 2 = the float delta the
// all-reduce left in dW (RCCL), 0 = nothing (peer exchange: its kernel applies the sum itself)
// RCCL (round 4): the ranks all-reduce the FIXED-POINT TABLE of the batch-step itself (kTabRep copies of A*F 64-bit integers, ncclInt64 /
// ncclSum, in place) and the next launch's prologue folds it exactly as it folds a single rank's own table (fold = 1): no table -> float
// kernel between the step and the collective (one dependent launch less per batch-step: 12.2 -> ~9.7 us at a size-1 communicator), and
// the sum over the ranks is an exact integer -- a run sharded in whole 512-learner blocks equals the unsharded run bit for bit, as it
// already did on the peer path.  fold = 2 (the float delta in dW) is no longer produced by the dense path.
static inline int fold_in_step(const rsrl_hip_ctx* c) { return !c->multi ? 1 : (c->cfg.exchange == RSRL_EXCHANGE_PEER ? 0 : 1); }
// the set of the rotating delta tables batch-step t accumulates into (models.hpp DeltaTab: t mod 3)
static inline long long* tab_set_of(const rsrl_hip_ctx* c, uint64_t t) { return c->sh_tab + (size_t)kTabRep * c->dw_elems * (size_t)(t % 3u); }
// the dense RCCL exchange: all-reduce of batch-step t's table set, in place.  Inside a captured graph t is the node's offset to a device-side
// counter that is a MULTIPLE OF 3 whenever a graph is replayed (train_now starts replaying only at such a step, graphs are 30 steps long),
// so t mod 3 is the set there too.
static int exchange_table(rsrl_hip_ctx* c, uint64_t t) {
    NCCL_TRY(ncclAllReduce(tab_set_of(c, t), tab_set_of(c, t), (size_t)kTabRep * c->dw_elems, ncclInt64, ncclSum, c->comm, c->stream));
    return RSRL_HIP_OK;
}
// dense basis, shared weights: ONE launch per batch-step (k_shared_step, models.hpp).  fold: add the previous batch-step's delta to
// the weights first.
static int enqueue_dense_step(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, DevStats* d_stats, int mode, int fold, uint64_t t,
                              const uint64_t* t_dev) {
    const float* W_in = c->sh_par ? c->W2 : c->W;
    float* W_out = fold ? (c->sh_par ? c->W : c->W2) : nullptr;
    bool ok = false;
    for_model(c, [&](auto tag) {
        using M = typename decltype(tag)::type;
        if constexpr (M::kDense) {
            hipLaunchKernelGGL((k_shared_step<M, kSharedBlock>), dim3(c->sh_rows), dim3(kSharedBlock), 0, c->stream, k, g, t, mode, W_in, W_out, c->sh_tab,
                               fold, c->dW, c->flags, d_stats, t_dev);
            ok = true;
        }
    });
    if (!ok) return NO_MODEL(c);
    KCHECK();
    if (fold) c->sh_par ^= 1;
    return RSRL_HIP_OK;
}
// xpart: 0 = the whole batch-step; 1 = everything BEFORE the RCCL all-reduce; 2 = what FOLLOWS it.  (1, 2: rsrl_hip_group_train issues
// the all-reduces of all ranks of a single-thread group between the two parts, inside one ncclGroupStart / End.)
static int enqueue_shared_step(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, DevStats* d_stats, int do_c, uint64_t t,
                               const uint64_t* t_dev, int xpart = 0) {
    const dim3 grid(grid_for(k.n_envs)), block(kBlock);
    const bool dense = c->cfg.basis == RSRL_FOURIER;
    if (dense) {
        if (xpart == 2) return RSRL_HIP_OK;                              // the next launch's prologue folds the all-reduced delta
        const int n = (int)c->dw_elems;
        const int fold = do_c ? fold_in_step(c) : 0;
        TRY(enqueue_dense_step(c, k, g, d_stats, (do_c ? 1 : 0) | 2, fold, t, t_dev));
        if (!c->multi) return RSRL_HIP_OK;
        // multi-rank: the delta table of this batch-step -> exchange; the sum reaches the weights in the exchange kernel (peer) or
        // in the next launch's prologue (RCCL: table -> dW -> all-reduce, folded as floats)
        if (c->cfg.exchange == RSRL_EXCHANGE_PEER) {                              // fused: delta -> every rank's slot; slots -> W
            hipLaunchKernelGGL(k_tab_exchange_apply, dim3(peer_grid(c, n)), dim3(256), 0, c->stream, c->sh_tab, n, k.alg.lr, c->d_peer_ptrs, c->peer_recv, c->W,
                               c->world_size, c->rank, t, t_dev, k.xdelta, c->d_peer_err, c->peer_timeout);
            KCHECK();
            return RSRL_HIP_OK;
        }
        if (xpart == 0) TRY(exchange_table(c, t));
        return RSRL_HIP_OK;
    }
    // SARSALambda / QLambda over the shared table (sparse per-learner traces, kernels_sparse_lambda.hpp) ride the same three launches: the step
    // kernel takes the TD target's residual (SARSA's / QLearning's formula, step size alpha), the scatter kernel is the one that also updates the traces
    const bool sparse_lambda = c->sp_keys != nullptr;
    const float step_size = (float)(sparse_lambda ? c->cfg.alpha : c->cfg.lr);
    if (xpart != 2 && !for_model(c, [&](auto tag) {
            using M = typename decltype(tag)::type;
            float* dwp = reinterpret_cast<float*>(c->dW_rep);
            const int nrep = c->n_rep;
            if constexpr (M::kSparse) {
                if (sparse_lambda) {
                    Common ks = k;
                    ks.alg.kind = c->cfg.algo == RSRL_SARSA_LAMBDA ? ALG_SARSA : ALG_QLEARNING; ks.alg.lr = step_size;
                    const int slice = (int)((int64_t)(c->F / c->cfg.n_tilings) * c->A);
                    hipLaunchKernelGGL((k_shared_ca<M>), grid, block, 0, c->stream, ks, g, t, do_c | (c->cfg.algo == RSRL_Q_LAMBDA ? 2 : 0), dwp, c->flags, d_stats, nrep,
                                       (int64_t)c->dw_elems, t_dev, c->sc_keys, c->sc_terms);
                    static const int per_env = getenv("RSRL_SPARSE_CHUNK") ? atoi(getenv("RSRL_SPARSE_CHUNK")) : 512;
                    const int per = per_env < 16 ? 16 : per_env;
                    const unsigned chunks = (unsigned)((k.n_envs + per - 1) / per);
                    const SparseTrace st{c->sp_keys, c->sp_vals, c->sp_len};
                    hipLaunchKernelGGL((k_sparse_trace_scatter<M::kT>), dim3(chunks, (unsigned)M::kT), dim3(1024), c->sp_lds ? (size_t)slice * 8 : 0, c->stream,
                                       c->sc_keys, c->sc_terms, c->flags, st, make_lambda(c), (int64_t)k.n_envs, (int64_t)k.n_envs, slice, per, c->dW_rep, nrep, (int64_t)c->dw_elems,
                                       FxScale(step_size).inv_lsb, c->sp_lds ? 1 : 0);
                    return;
                }
                if (c->sc_keys) {
                    // step kernel (terms + entries per learner) -> scatter kernel: block (chunk, tiling), 8 192 learners per chunk, one tiling's slice
                    // of the delta table (64-bit fixed-point accumulators) in LDS
                    const int slice = (int)((int64_t)(c->F / c->cfg.n_tilings) * c->A);
                    hipLaunchKernelGGL((k_shared_ca<M>), grid, block, 0, c->stream, k, g, t, do_c, dwp, c->flags, d_stats, nrep,
                                       (int64_t)c->dw_elems, t_dev, c->sc_keys, c->sc_terms);
                    static const int chunks_env = getenv("RSRL_SCATTER_CHUNKS") ? atoi(getenv("RSRL_SCATTER_CHUNKS")) : 32;
                    int64_t per = (k.n_envs + chunks_env - 1) / chunks_env;
                    per = ((per + 1023) / 1024) * 1024;
                    const unsigned chunks = (unsigned)((k.n_envs + per - 1) / per);
                    // (the apply folded into the scatter kernel -- its blocks meeting at a per-tiling arrival counter -- measured SLOWER than the third
                    // launch: 24.4 against 23.7 us per batch-step at 262 144 learners; scripts/ab/round6_pruned_knobs.patch)
                    hipLaunchKernelGGL(k_tile_scatter, dim3(chunks, (unsigned)c->cfg.n_tilings), dim3(1024), (size_t)slice * 8, c->stream, c->sc_keys, c->sc_terms,
                                       (int64_t)k.n_envs, slice, (int)per, c->dW_rep, nrep, (int64_t)c->dw_elems, FxScale((float)c->cfg.lr).inv_lsb);
                    return;
                }
            }
            hipLaunchKernelGGL((k_shared_ca<M>), grid, block, 0, c->stream, k, g, t, do_c, dwp, c->flags, d_stats, nrep, (int64_t)c->dw_elems, t_dev);
        })) return NO_MODEL(c);
    KCHECK();
    const int n = (int)c->dw_elems;
    const bool multi = c->multi;           // an exchange is attached: finalize -> exchange -> apply, also for a communicator of size 1
    if (xpart != 2) {
        hipLaunchKernelGGL(k_apply_rep, dim3(((n + 1) / 2 + 255) / 256), dim3(256), 0, c->stream, multi ? (float*)nullptr : c->W, c->dW, c->dW_rep, c->n_rep, n,
                           tile_lsb(step_size));
        KCHECK();
    }
    if (multi) {
        if (xpart == 0) TRY(exchange_dw(c, t, t_dev, k.xdelta));
        if (xpart != 1) {
            hipLaunchKernelGGL(k_apply_dw, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->W, c->dW, n);
            KCHECK();
        }
    }
    return RSRL_HIP_OK;
}
static int enqueue_shared_c(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, uint64_t t_last) {
    if (c->cfg.basis == RSRL_FOURIER) {
        // closing launch: fold the last batch-step's delta, phase C; the result goes back to the canonical buffer
        const int fold = fold_in_step(c);
        TRY(enqueue_dense_step(c, k, g, nullptr, 1, fold, t_last + 1, nullptr));
        if (c->sh_par) {
            HIP_TRY(hipMemcpyAsync(c->W, c->W2, c->w_bytes, hipMemcpyDeviceToDevice, c->stream));
            c->sh_par = 0;
        }
        return RSRL_HIP_OK;
    }
    if (!for_model(c, [&](auto tag) {
            using M = typename decltype(tag)::type;
            hipLaunchKernelGGL((k_shared_c<M>), dim3(grid_for(k.n_envs)), dim3(kBlock), 0, c->stream, k, g, t_last, c->flags);
        })) return NO_MODEL(c);
    KCHECK();
    return RSRL_HIP_OK;
}
// the single-step streaming kernel (register family, steps_per_launch = 1)
static int enqueue_k1_step(rsrl_hip_ctx* c, const Common& k, DevStats* d_stats, uint64_t t, const uint64_t* t_dev) {
    const dim3 gr(grid_for(k.n_envs)), b(kBlock);
    bool ok;
    const int kind = c->w_ls != 1 ? (c->k1_quad ? -3 : -2) : -1;              // learner-major rows: k_step_reg_q4 / k_step_reg_lm
    switch (c->cfg.domain) {
    case 0: ok = launch_train_reg_d0(c->cfg.order, c->cfg.algo, c->cfg.policy, gr, b, c->stream, k, t, kind, d_stats, t_dev); break;
    case 1: ok = launch_train_reg_d1(c->cfg.order, c->cfg.algo, c->cfg.policy, gr, b, c->stream, k, t, kind, d_stats, t_dev); break;
    default: ok = launch_train_reg_d2(c->cfg.order, c->cfg.algo, c->cfg.policy, gr, b, c->stream, k, t, kind, d_stats, t_dev); break;
    }
    if (!ok) return NO_MODEL(c);
    KCHECK();
    return RSRL_HIP_OK;
}

// ---- hipGraph replay of the launch-bound loops ------------------------------------------------------------------------
// One batch-step per launch costs ~4 us of launch gap per dependent kernel on top of the kernels themselves; kStepsPerGraph
// steady-state batch-steps (no statistics, single rank, ctx-owned stream) are captured once and replayed.  The nodes carry
// their step offset; the counter itself lives on the device (k_set_t before the first replay of a train call, k_advance_t
// as the graph's last node), so one executable graph serves every replay.  Any change of the kernel arguments (epsilon,
// pointers) re-captures.
constexpr int kStepsPerGraph = 32;
// the dense RCCL path all-reduces the table set of its batch-step (t mod 3): its graphs are 30 steps long and start at t = 0 mod 3
static inline bool rccl_dense(const rsrl_hip_ctx* c) { return c->multi && c->cfg.exchange == RSRL_EXCHANGE_RCCL && c->cfg.weight_mode == RSRL_W_SHARED && c->sh_tab != nullptr; }
static inline int steps_per_graph(const rsrl_hip_ctx* c) { return rccl_dense(c) ? 30 : kStepsPerGraph; }
static int ensure_step_graph(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, int kind) {
    if (c->step_graph_exec && c->step_graph_kind == kind && memcmp(&c->step_graph_key, &k, sizeof(Common)) == 0) return RSRL_HIP_OK;
    if (c->step_graph_exec) { (void)hipGraphExecDestroy(c->step_graph_exec); c->step_graph_exec = nullptr; }
    if (c->step_graph) { (void)hipGraphDestroy(c->step_graph); c->step_graph = nullptr; }
    HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = RSRL_HIP_OK;
    const int spg = steps_per_graph(c);
    for (int j = 0; j < spg && rc == RSRL_HIP_OK; ++j) {
        rc = kind == 1 ? enqueue_k1_step(c, k, nullptr, (uint64_t)j, c->d_t) : enqueue_shared_step(c, k, g, nullptr, 1, (uint64_t)j, c->d_t);
    }
    if (rc == RSRL_HIP_OK) hipLaunchKernelGGL(k_advance_t, dim3(1), dim3(1), 0, c->stream, c->d_t, (uint64_t)spg);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc != RSRL_HIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(RSRL_HIP_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    c->step_graph = graph;
    HIP_TRY(hipGraphInstantiate(&c->step_graph_exec, c->step_graph, nullptr, nullptr, 0));
    memcpy(&c->step_graph_key, &k, sizeof(Common));
    c->step_graph_kind = kind;
    return RSRL_HIP_OK;
}

// batch-steps per launch of the fused loops.  Every launch of the register-family loop loads and stores every learner's weights
// (60.7 MB at 65 536 MountainCar learners: ~11 us) and pays a launch-to-launch gap around its arithmetic (0.77 us per
// batch-step): 1 024 steps per launch instead of 256 is worth +7 % (8.3e10 -> 8.9e10 env-steps/s, 2 048: 9.0e10) and a
// launch still lasts under a millisecond (2.4 ms for the trace agents).  The memory-resident and wave-family loops keep 256
// (their steps are 15-150x longer).
static bool register_family_fused(const rsrl_hip_ctx* c) {
    const auto& g = c->cfg;
    return g.weight_mode == RSRL_W_PER_ENV && g.basis == RSRL_FOURIER && !is_wave(g) && !is_generic_fourier(g) && !has_aux(g.algo) &&
           !is_pred(g.algo) && g.algo != RSRL_Q_SIGMA;
}
static inline int64_t fuse_depth(const rsrl_hip_ctx* c) {
    const auto& g = c->cfg;
    if (g.steps_per_launch) return g.steps_per_launch;
    // every register-resident loop (also the trace / GreedyGQ / TD ones, which load and store two matrices per launch)
    const bool reg = g.weight_mode == RSRL_W_PER_ENV && g.basis == RSRL_FOURIER && !is_wave(g) && !is_generic_fourier(g) && g.algo != RSRL_Q_SIGMA;
    // round 3, under the driver's invocation (20-step calls, coalesced; scripts/gpu_r3_v6.sh): 1 024 -> 8.96e10, 2 048 -> 9.06e10,
    // 4 096 -> 9.12e10, 8 192 -> 9.17e10 env-steps/s; 4 096 (a 2.9 ms launch at 65 536 learners) is the default, RSRL_FUSE_DEPTH the A/B knob
    static const int64_t reg_depth = getenv("RSRL_FUSE_DEPTH") ? atoll(getenv("RSRL_FUSE_DEPTH")) : 4096;
    return reg ? (reg_depth > 0 ? reg_depth : 4096) : 256;
}

// ---- co-residency of the persistent kernel ----------------------------------------------------------------------------------
// Shared weights, dense basis: the whole train call as ONE persistent launch (kernels_persist.hpp).  k_shared_persist spins on
// granules written by the other blocks of its grid and by the grids of its peer ranks: every one of those blocks must be RESIDENT
// at the same time, or the resident ones wait for blocks that cannot start.  Three guards make that true by construction:
//  (1) the grid itself: sh_rows <= one 512-learner block per CU, provided the occupancy query admits at least one -- and, once per
//      ctx, ONE cooperative launch of the very same grid: the runtime's own check of the grid against that query (refused: the
//      per-step path takes over for good; a plain launch of the same grid has the same residency, so the later launches are plain);
//  (2) ranks of one peer group on one device: the SUM of their grids must fit.  Decided once and COLLECTIVELY in
//      rsrl_hip_peer_connect from what every rank put into its handle (rows, budgets, device identity, RSRL_NO_PERSIST): every
//      rank takes the same path -- the persistent and the per-step kernels exchange through different buffers and tags, so ranks
//      on different paths would never meet;
//  (3) unrelated ctxs of THIS process on one device: persist_admit() below -- one persistent group per device at a time (a lone
//      ctx that finds the device taken runs this call on the per-step path, which is bit-identical; a group waits on its stream).
// Persistent ctxs of OTHER processes that are not peers of this one cannot be seen from here: the bounded waits
// (config.peer_timeout_ms) are the backstop, and one process per GPU is the deployment.  RSRL_NO_PERSIST=1 keeps one launch per
// batch-step (k_shared_step), which is also what larger shards and RCCL-attached ctxs run.
static int persist_blocks_per_cu(rsrl_hip_ctx* c) {
    if (c->persist_occ >= 0) return c->persist_occ;
    int nb = 0;
    (void)hipSetDevice(c->cfg.device);
    for_model(c, [&](auto tag) {
        using M = typename decltype(tag)::type;
        if constexpr (M::kDense) {
            int a = 0, b = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(&k_shared_persist<M, kSharedBlock, true>), kSharedBlock, 0) != hipSuccess) a = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(&k_shared_persist<M, kSharedBlock, false>), kSharedBlock, 0) != hipSuccess) b = 0;
            nb = a < b ? a : b;
        }
    });
    (void)hipGetLastError();
    c->persist_occ = nb < 0 ? 0 : nb;
    return c->persist_occ;
}
// blocks of the persistent kernel the device may hold for ONE grid (a lone ctx, or one rank alone on its device): one per CU
static unsigned persist_budget_single(rsrl_hip_ctx* c) { return persist_blocks_per_cu(c) >= 1 ? (unsigned)c->n_cu : 0u; }
// ... and for the SUM of the grids of several ranks on one device.  The occupancy query over-reports by one block per CU for some
// kernels on this runtime (MI355X_MICROARCH.md, "Residency and cooperative launch"), so one block per CU is held back
static unsigned persist_budget_shared(rsrl_hip_ctx* c) {
    const int nb = persist_blocks_per_cu(c);
    return nb >= 1 ? (unsigned)c->n_cu * (unsigned)(nb > 1 ? nb - 1 : 1) : 0u;
}
// this rank alone: could it run the persistent kernel?  (multi-rank: what goes into the handle; the group decides)
static bool persist_capable(rsrl_hip_ctx* c) {
    if (c->cfg.weight_mode != RSRL_W_SHARED || !c->sh_tab) return false;
    if (getenv("RSRL_NO_PERSIST")) return false;
    return c->sh_rows <= persist_budget_single(c);
}
static bool persist_ok(rsrl_hip_ctx* c) {
    if (c->cfg.weight_mode != RSRL_W_SHARED || !c->sh_tab || c->persist_refused) return false;
    if (c->multi) return c->cfg.exchange == RSRL_EXCHANGE_PEER && c->group_persist;      // the GROUP's decision, never this rank's own
    return persist_capable(c);
}
static int ensure_persist_buffers(rsrl_hip_ctx* c) {
    const size_t pairs = (c->dw_elems + 1) / 2;
    if (!c->px_A) {
        const size_t bytes = sizeof(unsigned long long) * pairs * c->sh_rows * 2;
        HIP_TRY(hipMalloc((void**)&c->px_A, bytes));
        HIP_TRY(hipMemsetAsync(c->px_A, 0, bytes, c->stream));          // tag 0 never matches
    }
    if (!c->px_B) {                                                     // single rank: a private hop-2 buffer
        const size_t bytes = sizeof(unsigned long long) * 2 * pairs * 2;
        HIP_TRY(hipMalloc((void**)&c->px_B, bytes));
        c->px_B_owned = true;
        HIP_TRY(hipMemsetAsync(c->px_B, 0, bytes, c->stream));
        HIP_TRY(hipMalloc((void**)&c->d_px_Bptrs, sizeof(void*)));
        HIP_TRY(hipMemcpyAsync(c->d_px_Bptrs, &c->px_B, sizeof(void*), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));                       // &c->px_B is read by the copy
    }
    return RSRL_HIP_OK;
}

// (3) one persistent group per device at a time, within this process.  An entry = one persistent launch in flight: who owns it (a
// lone ctx, or a peer group by its token), which round of the group it belongs to (the exchange sequence number at its start: the
// ranks of a group agree on it), and which foreign launches its stream was made to wait for.  The ranks of one round must make
// the SAME waits: a rank that started while a foreign grid still held CUs, spinning for a peer that waits for that grid to end,
// could keep the foreign grid from ever becoming resident.
namespace {
struct PersistEntry { uint64_t id; hipEvent_t ev; uint64_t owner; uint64_t round; std::vector<uint64_t> waited; };
struct PersistGate { std::mutex mu; uint64_t next_id = 1; std::map<int, std::vector<PersistEntry>> by_dev; };
PersistGate* persist_gate() { static PersistGate* g = new PersistGate(); return g; }      // never destroyed: ctxs may outlive static destructors
}
enum { PERSIST_LAUNCHED = 0, PERSIST_FALLBACK = 1 };
// Launch one chunk of the persistent kernel under the gate.  *outcome = PERSIST_FALLBACK: nothing was launched, the caller (a lone
// ctx only) runs this call on the per-step path.
static int persist_launch(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, int64_t n_steps, DevStats* d_stats, bool may_fall_back, int* outcome) {
    *outcome = PERSIST_LAUNCHED;
    TRY(ensure_persist_buffers(c));
    PersistExch x{};
    x.A = c->px_A; x.B = c->d_px_Bptrs; x.B_self = c->px_B; x.err = c->d_peer_err;
    x.world = c->multi ? c->world_size : 1; x.rank = c->multi ? c->rank : 0; x.timeout_ticks = c->peer_timeout;
    PersistGate& gate = *persist_gate();
    std::lock_guard<std::mutex> lock(gate.mu);                          // admission, launch and registration are one step
    std::vector<PersistEntry>& live = gate.by_dev[c->cfg.device];
    for (auto it = live.begin(); it != live.end();) {                   // launches that have ended leave the gate
        if (hipEventQuery(it->ev) == hipSuccess) { (void)hipEventDestroy(it->ev); it = live.erase(it); }
        else { (void)hipGetLastError(); ++it; }
    }
    const uint64_t owner = c->group_token ? c->group_token : (uint64_t)(uintptr_t)c;
    const uint64_t round = c->px_seq;
    std::vector<uint64_t> waited;
    auto wait_for = [&](uint64_t id) -> int {
        for (const PersistEntry& e : live)
            if (e.id == id) { HIP_TRY(hipStreamWaitEvent(c->stream, e.ev, 0)); waited.push_back(id); }
        return RSRL_HIP_OK;                                             // (an entry that has left the gate has ended: nothing to wait for)
    };
    const PersistEntry* opener = nullptr;
    if (c->group_token)
        for (const PersistEntry& e : live) if (e.owner == owner && e.round == round) { opener = &e; break; }
    if (opener) {                                                       // a peer of this round is already in: make its waits, nothing else
        const std::vector<uint64_t> ids = opener->waited;
        for (uint64_t id : ids) TRY(wait_for(id));
    } else {
        std::vector<uint64_t> foreign;
        for (const PersistEntry& e : live) if (e.owner != owner) foreign.push_back(e.id);
        if (!foreign.empty() && may_fall_back) { *outcome = PERSIST_FALLBACK; return RSRL_HIP_OK; }
        for (uint64_t id : foreign) TRY(wait_for(id));
    }
    // (1) the runtime's own check of this grid, once per ctx: a cooperative launch (+15-19 us of host time, paid once).  Not when a
    // peer rank shares this process AND device: cooperative launches of one process go through one queue per device, and a rank
    // queued behind the peer it exchanges with would wait for itself.
    static const bool no_coop = getenv("RSRL_PERSIST_NO_COOP") != nullptr;
    const bool coop = !c->coop_validated && c->coop_allowed && !no_coop;
    bool ok = false;
    hipError_t coop_err = hipSuccess;
    for_model(c, [&](auto tag) {
        using M = typename decltype(tag)::type;
        if constexpr (M::kDense) {
            Common kk = k; BasisGeom gg = g; uint64_t t0 = c->t, xs0 = c->px_seq; int n = (int)n_steps; float* W = c->W; PersistExch xx = x; DevStats* st = d_stats;
            void* args[] = {&kk, &gg, &t0, &xs0, &n, &W, &xx, &st};
            if (coop) {
                const void* fn = c->multi ? reinterpret_cast<const void*>(&k_shared_persist<M, kSharedBlock, true>)
                                          : reinterpret_cast<const void*>(&k_shared_persist<M, kSharedBlock, false>);
                coop_err = hipLaunchCooperativeKernel(fn, dim3(c->sh_rows), dim3(kSharedBlock), args, 0, c->stream);
            } else if (c->multi) {
                hipLaunchKernelGGL((k_shared_persist<M, kSharedBlock, true>), dim3(c->sh_rows), dim3(kSharedBlock), 0, c->stream, kk, gg, t0, xs0, n, W, xx, st);
            } else {
                hipLaunchKernelGGL((k_shared_persist<M, kSharedBlock, false>), dim3(c->sh_rows), dim3(kSharedBlock), 0, c->stream, kk, gg, t0, xs0, n, W, xx, st);
            }
            ok = true;
        }
    });
    if (!ok) return NO_MODEL(c);
    if (coop) {
        if (coop_err != hipSuccess) {
            (void)hipGetLastError();
            if (may_fall_back) { c->persist_refused = true; *outcome = PERSIST_FALLBACK; return RSRL_HIP_OK; }
            return fail(RSRL_HIP_ERCCL, "the runtime refused the persistent shared-W grid of rank %d (%u blocks of %d threads on device %d: %s); the other ranks "
                                        "of the group were told it fits -- set RSRL_NO_PERSIST=1 on every rank", c->rank, c->sh_rows, kSharedBlock, c->cfg.device,
                        hipGetErrorString(coop_err));
        }
        c->coop_validated = true;
    }
    KCHECK();
    PersistEntry e;
    e.id = gate.next_id++; e.owner = owner; e.round = round; e.waited = waited;
    HIP_TRY(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e.ev, c->stream));
    live.push_back(std::move(e));
    return RSRL_HIP_OK;
}

// SARSALambda / QLambda over one shared tile table (kernels_sparse_lambda.hpp): per batch-step phase A (one wave per learner: residual against W_t,
// sparse trace update, the learner's terms into the fixed-point table), the table -> W (the same finalize -> [exchange] -> apply as rsrl_hip_handle),
// phase C (sample from W_{t+1}, restarts).  Plain launches: correctness first.
// One batch-step of SARSALambda / QLambda over a shared tile table in two halves around the exchange of the delta (rsrl_hip_group_train's RCCL branch
// runs the halves of all its ranks in lock-step with the all-reduces grouped between them):
//   A: every learner's step (transition, TD error, sparse trace, fixed-point scatter of alpha * residual * z) + the table -> float delta
//   B: W += delta (summed over the ranks by then), the behaviour policy's sample with the updated table
static int train_now(rsrl_hip_ctx* c, int64_t n_steps, rsrl_hip_stats* stats_out) {
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->tq_valid = false;                 // the weights move behind the trait path's hand-over cache
    DevStats* d_stats = stats_out ? c->d_stats : nullptr;      // statistics cost a block reduction per launch: opt-in
    if (d_stats) HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(DevStats) * c->n_stat_slots, c->stream));
    Common k = make_common(c);
    const BasisGeom g = make_geom(c);
    const bool shared = c->cfg.weight_mode == RSRL_W_SHARED;
    const bool fourier = c->cfg.basis == RSRL_FOURIER;
    const int64_t spl = shared ? 1 : fuse_depth(c);
    // single-step streaming kernel: needs the whole W addressable through one 32-bit buffer descriptor
    const bool stream_k1 = !shared && fourier && !is_wave(c->cfg) && !is_generic_fourier(c->cfg) && !has_aux(c->cfg.algo) && !is_pred(c->cfg.algo) && c->cfg.algo != RSRL_Q_SIGMA &&
                           spl == 1 && (uint64_t)c->w_elems * 4ull < (1ull << 32);
    // launch-bound loops go through a captured graph (RSRL_NO_GRAPH=1 keeps the plain launches, for A/B runs)
    // (multi-rank included: the RCCL all-reduce and the peer-exchange kernels are captured with the step like any other node)
    const bool graph_ok = (stream_k1 || shared) && c->own_stream && !stats_out && !getenv("RSRL_NO_GRAPH");
    bool t_dev_set = false;
    int64_t done = 0;
    // the delta tables rotate with the batch-step counter: a counter that did not simply continue (reset, restored checkpoint)
    // finds them in another phase -- start from clean tables then
    bool persist = shared && n_steps > 0 && persist_ok(c);
    if (persist) {
        for (int64_t left = n_steps; left > 0;) {                       // (the kernel's step count is an int)
            const int64_t chunk = left < (int64_t)1 << 30 ? left : (int64_t)1 << 30;
            TRY(timing_begin(c));
            int outcome = PERSIST_LAUNCHED;
            // only a LONE ctx may change its mind here (its two paths are bit-identical and self-contained), and only before its first chunk
            TRY(persist_launch(c, k, g, chunk, d_stats, !c->multi && left == n_steps, &outcome));
            if (outcome == PERSIST_FALLBACK) { persist = false; break; }
            TRY(timing_end(c, (uint32_t)chunk));
            c->t += (uint64_t)chunk; c->px_seq += (uint64_t)chunk; left -= chunk;
            k = make_common(c);
        }
        if (persist) { c->kernel_name = "k_shared_persist"; done = n_steps; }
    }
    const bool peer_steps = shared && c->multi && c->cfg.exchange == RSRL_EXCHANGE_PEER && !persist;   // per-step exchanges on peer_recv
    if (shared && !persist && c->sh_tab && n_steps > 0 && c->t != c->sh_tab_t)
        HIP_TRY(hipMemsetAsync(c->sh_tab, 0, sizeof(long long) * 3 * kTabRep * c->dw_elems, c->stream));
    while (done < n_steps) {
        // (dense shared W: the W / row buffers alternate every batch-step, the graph is captured at the parity of an odd step count)
        const int spg = steps_per_graph(c);
        if (graph_ok && n_steps - done >= spg && (shared ? (done > 0 && (!fourier || (done & 1)) && (!rccl_dense(c) || c->t % 3u == 0)) : c->q_valid)) {
            // the graph's nodes read the policy parameters from device memory: set_epsilon between calls (the reference's drivers
            // decay epsilon every episode, examples/sarsa_lambda.rs:68) refreshes 48 bytes instead of re-instantiating 32+ nodes
            Common kg = k; kg.q_valid = stream_k1 ? 1 : k.q_valid;
            kg.dyn = c->d_dyn; kg.pol = PolicyParams{}; kg.apol = PolicyParams{};
            const DynParams want{k.pol, k.apol};
            if (!c->dyn_valid || memcmp(&want, &c->dyn_uploaded, sizeof(want)) != 0) {
                hipLaunchKernelGGL(k_set_dyn, dim3(1), dim3(1), 0, c->stream, c->d_dyn, want); KCHECK();
                c->dyn_uploaded = want; c->dyn_valid = true;
            }
            TRY(ensure_step_graph(c, kg, g, stream_k1 ? 1 : 2));
            if (!t_dev_set) { hipLaunchKernelGGL(k_set_t, dim3(1), dim3(1), 0, c->stream, c->d_t, c->t); KCHECK(); t_dev_set = true; }
            TRY(timing_begin(c));
            HIP_TRY(hipGraphLaunch(c->step_graph_exec, c->stream));
            TRY(timing_end(c, (uint32_t)spg));
            c->kernel_name = stream_k1 ? (c->w_ls != 1 ? (c->k1_quad ? "k_step_reg_q4" : "k_step_reg_lm") : "k_step_reg") : shared_kernel_name(c);
            c->t += (uint64_t)spg;
            if (peer_steps) c->peer_seq += (uint64_t)spg;
            done += spg;
            continue;
        }
        t_dev_set = false;                  // plain launches advance the host counter only
        const int chunk = (int)((n_steps - done < spl) ? (n_steps - done) : spl);
        TRY(timing_begin(c));
        if (shared) {
            TRY(enqueue_shared_step(c, k, g, d_stats, done == 0 ? 0 : 1, c->t, nullptr));
            c->kernel_name = shared_kernel_name(c);
        } else if (is_wave(c->cfg) && is_wave_aux_algo(c->cfg.algo)) {
            launch_wave_aux(c, dim3(wave_grid_for(k.n_envs)), k, make_wave_aux(c), c->t, chunk, d_stats, (const float*)nullptr, (const int32_t*)nullptr,
                            (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, (int64_t)0, (float*)nullptr);
            c->kernel_name = "k_wave_aux";
            KCHECK();
        } else if (is_pred(c->cfg.algo) && c->cfg.basis == RSRL_TILE_CODING) {
            if (!launch_td_tile(c->cfg.domain, c->cfg.n_tilings, c->cfg.algo == RSRL_TD_LAMBDA, k.n_envs, c->stream, k, g, make_td(c), c->t, chunk, d_stats,
                                nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr)) return NO_MODEL(c);
            c->kernel_name = "k_td_tile";
            KCHECK();
        } else if (is_pred(c->cfg.algo) && is_generic_fourier(c->cfg)) {
            if (!launch_td_model(c->cfg, dim3(grid_for(k.n_envs)), dim3(kBlock), c->stream, k, make_td(c), g, c->cfg.algo == RSRL_TD_LAMBDA, c->t, chunk, d_stats,
                                 nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr)) return NO_MODEL(c);
            c->kernel_name = "k_td_mem";
            KCHECK();
        } else if (is_pred(c->cfg.algo)) {
            if (!launch_train_td(c->cfg.domain, c->cfg.order, c->cfg.algo == RSRL_TD_LAMBDA, dim3(grid_for(k.n_envs)), dim3(kBlock), c->stream, k,
                                 make_td(c), c->t, chunk, d_stats)) return NO_MODEL(c);
            c->kernel_name = "k_train_td";
            KCHECK();
        } else if (c->cfg.algo == RSRL_Q_SIGMA && is_wave(c->cfg)) {
            if (c->cfg.domain == RSRL_CART_POLE) hipLaunchKernelGGL((k_wave_qsigma<1>), dim3(wave_grid_for(k.n_envs)), dim3(kBlock), 0, c->stream, k, make_qs(c), c->t, chunk, d_stats, (const float*)nullptr, (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, (int64_t)0, (float*)nullptr);
            else hipLaunchKernelGGL((k_wave_qsigma<2>), dim3(wave_grid_for(k.n_envs)), dim3(kBlock), 0, c->stream, k, make_qs(c), c->t, chunk, d_stats, (const float*)nullptr, (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, (int64_t)0, (float*)nullptr);
            c->kernel_name = "k_wave_qsigma";
            KCHECK();
        } else if (c->cfg.algo == RSRL_Q_SIGMA) {
            const bool reg = fourier && !is_generic_fourier(c->cfg);
            if (!(reg ? launch_qsigma(c->cfg.domain, c->cfg.order, dim3(grid_for(k.n_envs)), dim3(kBlock), c->stream, k, make_qs(c), g, c->t, chunk, d_stats,
                                      nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr)
                      : launch_qsigma_model(c->cfg, dim3(grid_for(k.n_envs)), dim3(kBlock), c->stream, k, make_qs(c), g, c->t, chunk, d_stats,
                                            nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr))) return NO_MODEL(c);
            c->kernel_name = "k_train_qsigma";
            KCHECK();
        } else if (c->cfg.algo == RSRL_GREEDY_GQ) {
            const bool reg = fourier && !is_generic_fourier(c->cfg);
            if (!(reg ? launch_train_gq(c->cfg.domain, c->cfg.order, c->cfg.policy, dim3(grid_for(k.n_envs)), dim3(kBlock), c->stream, k,
                                        make_gq(c), c->t, chunk, d_stats)
                      : launch_gq_model(c->cfg, dim3(grid_for(k.n_envs)), dim3(kBlock), c->stream, k, make_gq(c), g, c->t, chunk, d_stats,
                                        nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr))) return NO_MODEL(c);
            c->kernel_name = reg ? "k_train_gq" : "k_train_gq_mem";
            KCHECK();
        } else if (is_lambda(c->cfg.algo) && c->cfg.basis == RSRL_TILE_CODING) {
            if (!launch_lambda_tile(c->cfg.domain, c->cfg.n_tilings, k.n_envs, c->stream, k, g, make_lambda(c), c->t, chunk, d_stats, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, 0, nullptr)) return NO_MODEL(c);
            c->kernel_name = "k_lambda_tile";
            KCHECK();
        } else if (is_lambda(c->cfg.algo) && is_wave(c->cfg)) {
            for_wave(c, [&](auto tag) {
                using T = decltype(tag); using WT = typename T::wt;
                hipLaunchKernelGGL((k_wave_lambda<T::domain, WT>), dim3(wave_grid_for(k.n_envs)), dim3(kBlock), 0, c->stream, k, make_lambda(c), (WT*)c->W, c->t, chunk, d_stats,
                                   (const float*)nullptr, (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, (int64_t)0, (float*)nullptr);
            });
            c->kernel_name = "k_wave_lambda";
            KCHECK();
        } else if (is_lambda(c->cfg.algo) && is_generic_fourier(c->cfg)) {
            if (!launch_lambda_model(c->cfg, dim3(grid_for(k.n_envs)), dim3(kBlock), c->stream, k, make_lambda(c), g, c->t, chunk, d_stats, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, 0, nullptr)) return NO_MODEL(c);
            c->kernel_name = "k_train_lambda_mem";
            KCHECK();
        } else if (is_lambda(c->cfg.algo)) {
            if (!launch_train_lambda(c->cfg.domain, c->cfg.order, c->cfg.algo, c->cfg.policy, dim3(grid_for(k.n_envs)), dim3(kBlock),
                                     c->stream, k, make_lambda(c), c->t, chunk, d_stats)) return NO_MODEL(c);
            c->kernel_name = "k_train_lambda";
            KCHECK();
        } else if (is_wave(c->cfg)) {
            // bf16 weights: the packed-register kernel, two waves per SIMD (kernels_wave.hpp; RSRL_WAVE_PK=0 keeps the fp32-register one: A/B, same bits)
            static const bool wave_pk = !(getenv("RSRL_WAVE_PK") && getenv("RSRL_WAVE_PK")[0] == '0');
            const bool pk = wave_pk && c->cfg.weight_dtype == RSRL_W_BF16;
            for_wave(c, [&](auto tag) {
                using T = decltype(tag); using WT = typename T::wt;
                const dim3 wg(wave_grid_for(k.n_envs)), wb(kBlock);
                if constexpr (WaveIO<WT>::kBf16) {
                    if (pk) {
                        if (k.eps) hipLaunchKernelGGL((k_train_wave_pk<T::domain, true>), wg, wb, 0, c->stream, k, (WT*)c->W, c->t, chunk, d_stats);      // the per-learner epsilon schedule
                        else hipLaunchKernelGGL((k_train_wave_pk<T::domain>), wg, wb, 0, c->stream, k, (WT*)c->W, c->t, chunk, d_stats);
                        return;
                    }
                }
                if (k.eps) hipLaunchKernelGGL((k_train_wave<T::domain, WT, true>), wg, wb, 0, c->stream, k, (WT*)c->W, c->t, chunk, d_stats);
                else hipLaunchKernelGGL((k_train_wave<T::domain, WT>), wg, wb, 0, c->stream, k, (WT*)c->W, c->t, chunk, d_stats);
            });
            c->kernel_name = pk ? "k_train_wave_pk" : "k_train_wave";
            KCHECK();
        } else if (stream_k1) {
            TRY(enqueue_k1_step(c, k, d_stats, c->t, nullptr));
            c->kernel_name = c->w_ls != 1 ? (c->k1_quad ? "k_step_reg_q4" : "k_step_reg_lm") : "k_step_reg";
            c->q_valid = true; k.q_valid = 1;
        } else if (fourier && !is_generic_fourier(c->cfg)) {
            const dim3 gr(grid_for(k.n_envs)), b(kBlock);
            const int kchunk = chunk;
            bool ok;
            switch (c->cfg.domain) {
            case 0: ok = launch_train_reg_d0(c->cfg.order, c->cfg.algo, c->cfg.policy, gr, b, c->stream, k, c->t, kchunk, d_stats); break;
            case 1: ok = launch_train_reg_d1(c->cfg.order, c->cfg.algo, c->cfg.policy, gr, b, c->stream, k, c->t, kchunk, d_stats); break;
            default: ok = launch_train_reg_d2(c->cfg.order, c->cfg.algo, c->cfg.policy, gr, b, c->stream, k, c->t, kchunk, d_stats); break;
            }
            if (!ok) return NO_MODEL(c);
            c->kernel_name = "k_train_reg";
            KCHECK();
            c->q_valid = true; k.q_valid = 1;       // the launch left Q(s,.) of its final state in qcache
        } else {
            if (!for_model(c, [&](auto tag) {
                    using M = typename decltype(tag)::type;
                    hipLaunchKernelGGL((k_train_mem<M>), dim3(grid_for(k.n_envs)), dim3(kBlock), 0, c->stream, k, g, c->t, chunk, d_stats);
                })) return NO_MODEL(c);
            c->kernel_name = "k_train_mem";
            KCHECK();
            c->q_valid = false;
        }
        TRY(timing_end(c));
        c->t += (uint64_t)chunk;
        if (peer_steps) c->peer_seq += (uint64_t)chunk;
        done += chunk;
    }
    if (shared && n_steps > 0 && !persist) { TRY(enqueue_shared_c(c, k, g, c->t - 1)); c->sh_tab_t = c->t; }     // phase C of the last batch-step
    if (stats_out) {
        HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(DevStats) * c->n_stat_slots, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        memset(stats_out, 0, sizeof(*stats_out));
        stats_out->env_steps = (uint64_t)n_steps * (uint64_t)c->cfg.n_envs;
        for (size_t b = 0; b < c->n_stat_slots; ++b) {          // fixed order: reproducible sums
            stats_out->episodes += c->h_stats[b].episodes;
            stats_out->episodes_truncated += c->h_stats[b].episodes_truncated;
            stats_out->sum_episode_steps += c->h_stats[b].sum_episode_steps;
            stats_out->sum_abs_td_error += c->h_stats[b].sum_abs_td_error;
            stats_out->sum_reward += c->h_stats[b].sum_reward;
        }
    }
    return RSRL_HIP_OK;
}

// ---- the trait-granular loop (kernels_trait.hpp) --------------------------------------------------------------------------------------
static inline bool trait_fast(const rsrl_hip_ctx* c) { return c->tq_key != nullptr; }
// the hand-over cache is keyed by the state an entry belongs to; whenever the weights changed behind it (train, set / load weights) the keys are
// emptied (NaN: never equal to a state) before the next kernel that looks at them
static int trait_cache_ready(rsrl_hip_ctx* c) {
    if (c->tq_valid) return RSRL_HIP_OK;
    HIP_TRY(hipMemsetAsync(c->tq_key, 0xFF, sizeof(float) * (size_t)c->D * (size_t)c->cfg.n_envs, c->stream));
    c->tq_valid = true;
    return RSRL_HIP_OK;
}
static int launch_domain_step(rsrl_hip_ctx* c, const Common& k, const int32_t* d_act, float* from, float* next, float* rew, uint8_t* term) {
    const dim3 g(grid_for(c->cfg.n_envs)), b(kBlock);
    switch (c->cfg.domain) {
    case 0: hipLaunchKernelGGL(k_domain_step<0>, g, b, 0, c->stream, k, d_act, from, next, rew, term); break;
    case 1: hipLaunchKernelGGL(k_domain_step<1>, g, b, 0, c->stream, k, d_act, from, next, rew, term); break;
    default: hipLaunchKernelGGL(k_domain_step<2>, g, b, 0, c->stream, k, d_act, from, next, rew, term); break;
    }
    KCHECK();
    return RSRL_HIP_OK;
}
static int launch_domain_reset(rsrl_hip_ctx* c, const Common& k, const uint8_t* d_mask) {
    const dim3 g(grid_for(c->cfg.n_envs)), b(kBlock);
    switch (c->cfg.domain) {
    case 0: hipLaunchKernelGGL(k_domain_reset<0>, g, b, 0, c->stream, k, d_mask); break;
    case 1: hipLaunchKernelGGL(k_domain_reset<1>, g, b, 0, c->stream, k, d_mask); break;
    default: hipLaunchKernelGGL(k_domain_reset<2>, g, b, 0, c->stream, k, d_mask); break;
    }
    KCHECK();
    return RSRL_HIP_OK;
}
// Handler::handle on the fast path: one pass over the learners' weight images, the hand-over left for the sample that follows
static int launch_trait_handle(rsrl_hip_ctx* c, const Common& k, const float* from, const int32_t* act, const float* rew, const float* to,
                               const uint8_t* term, int64_t M, uint64_t t, float* td) {
    TRY(trait_cache_ready(c));
    TraitIo io{};
    io.from = from; io.act = act; io.rew = rew; io.to = to; io.termf = term; io.td_out = td; io.qkey = c->tq_key; io.Mn = M;
    TRY(timing_begin(c));
    if (!launch_trait_lm(c->cfg.domain, c->cfg.order, c->cfg.algo, -1, c->stream, k, io, t)) return NO_MODEL(c);
    KCHECK();
    c->kernel_name = "k_trait_lm<handle>";
    return timing_end(c);
}
// the deferred calls, one kernel per call, in the order they were made
static int trait_flush(rsrl_hip_ctx* c) {
    if (c->tp.stage == 0) return RSRL_HIP_OK;
    const rsrl_hip_ctx::TraitPend p = c->tp;
    c->tp.stage = 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    const Common k = make_common(c);
    TRY(launch_domain_step(c, k, p.act, p.from, p.to, p.rew, p.term));
    if (p.stage >= 2) TRY(launch_trait_handle(c, k, p.from, p.act, p.rew, p.to, p.term, c->cfg.n_envs, p.t_handle, p.td));
    if (p.stage >= 3) TRY(launch_domain_reset(c, k, p.term));
    return RSRL_HIP_OK;
}

static int flush_pending(rsrl_hip_ctx* c) {
    if (!c) return RSRL_HIP_OK;
    if (c->tp.stage) TRY(trait_flush(c));
    if (c->pending == 0) return RSRL_HIP_OK;
    const int64_t n = c->pending;
    c->pending = 0;
    return train_now(c, n, nullptr);
}
// fused register-family loop: any split of n batch-steps into launches gives bit-identical results (Q(s,.) is carried between
// launches, the RNG is addressed by the batch-step) -- the property launch coalescing relies on (tests: fused == stepwise)
// Only on a ctx-OWNED stream: a caller who supplied config.stream orders its own work on it (hipStreamSynchronize, events, a
// capture in progress -- which hipStreamQuery would invalidate); everything train() accepted must be on that stream when it returns.
static bool coalescable(const rsrl_hip_ctx* c) {
    return c->own_stream && register_family_fused(c) && c->cfg.steps_per_launch != 1 && !getenv("RSRL_NO_COALESCE");
}
// rsrl_hip_train is asynchronous when no statistics are requested: it returns once the work is accepted.  A short call (the
// 20 batch-steps of a driver loop) costs a full load + store of every learner's weights around ~20 us of arithmetic, so calls
// that arrive while the stream is still busy are COALESCED: their steps are held back and launched fuse-depth (4 096) at a time,
// or as soon as anything observes or changes the ctx (every other entry point flushes first, rsrl_hip_sync included), or
// when a call finds the stream idle (then nothing is gained by waiting).  Invisible to the caller: same results bit for bit,
// same ordering; 5 000 back-to-back train(20) calls run as ~400 launches instead of 5 000.  RSRL_NO_COALESCE=1 disables it.
#define ST_RCCL_GUARD(c) do { if ((c)->st_rccl_group) return fail(RSRL_HIP_ESTATE, "this ctx is a rank of a single-thread RCCL group: its collectives must be " \
    "issued for all ranks together -- step the group with rsrl_hip_group_train (or use one thread / process per rank, or RSRL_EXCHANGE_PEER)"); } while (0)
int rsrl_hip_train(rsrl_hip_ctx* c, int64_t n_steps, rsrl_hip_stats* stats_out) {
    CHECK_CTX(c);
    if (n_steps < 0) return fail(RSRL_HIP_EINVAL, "n_steps < 0");
    ST_RCCL_GUARD(c);
    if (stats_out || !coalescable(c)) {
        FLUSH(c);
        return train_now(c, n_steps, stats_out);
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->pending += n_steps;
    const int64_t depth = fuse_depth(c);
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess) return flush_pending(c);                       // idle stream: launch now
    if (q != hipErrorNotReady) return fail(RSRL_HIP_EHIP, "hipStreamQuery: %s", hipGetErrorString(q));
    (void)hipGetLastError();
    if (c->pending >= depth) {
        const int64_t n = c->pending - c->pending % depth;
        c->pending -= n;
        return train_now(c, n, nullptr);
    }
    return RSRL_HIP_OK;
}

static int rollout_impl(rsrl_hip_ctx* c, int64_t step_limit, int64_t M, uint32_t* n_states_out, float* total_reward_out, float* states_out,
                        int32_t* actions_out, float* rewards_out, uint8_t* terminal_out, const RolloutPolicy& rp) {
    CHECK_CTX(c); FLUSH(c);
    if (!n_states_out) return fail(RSRL_HIP_EINVAL, "null argument");
    if (step_limit == 0) {
        // Domain::rollout(.., None) (rsrl_domains/src/lib.rs:469-476 collects until the first Terminal observation, without a limit).  A device
        // loop needs a bound: the ctx's max_episode_steps, the same cap the driver loop truncates episodes at -- at most that many transitions,
        // so a trajectory that terminates within the cap is exactly the reference's unbounded one
        if (c->cfg.max_episode_steps == 0) return fail(RSRL_HIP_EINVAL, "step_limit 0 (no limit, Domain::rollout(.., None)) needs config.max_episode_steps > 0 as the bound");
        step_limit = (int64_t)c->cfg.max_episode_steps + 1;
    }
    if (step_limit < 1) return fail(RSRL_HIP_EINVAL, "step_limit must be >= 1, or 0 for no limit (bounded by config.max_episode_steps)");
    if (M < 1 || M > c->cfg.n_envs) return fail(RSRL_HIP_EINVAL, "bad batch (M=%lld, n_envs=%lld)", (long long)M, (long long)c->cfg.n_envs);
    if (is_pred(c->cfg.algo)) return fail(RSRL_HIP_ESTATE, "a prediction agent has a state-value function only: no action values to roll out with");
    if (!rp.sample && c->cfg.policy == RSRL_RANDOM) return fail(RSRL_HIP_EINVAL, "Random policy has no mode.");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (step_limit == 1) { actions_out = nullptr; rewards_out = nullptr; }      // Trajectory.steps is empty
    OutBuf<uint32_t> on; OutBuf<float> ot, os, orw; OutBuf<int32_t> oa; OutBuf<uint8_t> otm;
    const size_t tr_rows = (size_t)(step_limit - 1) * (size_t)M;
    TRY(stage_out(c, 0, n_states_out, (size_t)M, &on));
    TRY(stage_out(c, 1, total_reward_out, (size_t)M, &ot));
    TRY(stage_out(c, 2, states_out, (size_t)step_limit * c->D * (size_t)M, &os));
    TRY(stage_out(c, 3, actions_out, tr_rows, &oa));
    TRY(stage_out(c, 4, rewards_out, tr_rows, &orw));
    TRY(stage_out(c, 5, terminal_out, (size_t)M, &otm));
    // rows past a trajectory's end stay as the caller left them in device memory; staged host outputs start from zero
    if (os.staged) HIP_TRY(hipMemsetAsync(os.dev, 0, sizeof(float) * os.count, c->stream));
    if (oa.staged) HIP_TRY(hipMemsetAsync(oa.dev, 0, sizeof(int32_t) * oa.count, c->stream));
    if (orw.staged) HIP_TRY(hipMemsetAsync(orw.dev, 0, sizeof(float) * orw.count, c->stream));
    const TrajOut tr{os.dev, oa.dev, orw.dev, otm.dev, M};
    const Common k = make_common(c);
    const BasisGeom g = make_geom(c);
    if (is_wave(c->cfg)) {
        for_wave(c, [&](auto tag) {
            using T = decltype(tag); using WT = typename T::wt;
            hipLaunchKernelGGL((k_wave_rollout<T::domain, WT>), dim3(wave_grid_for(M)), dim3(kBlock), 0, c->stream, k, (const WT*)c->W, step_limit, on.dev, ot.dev, M, tr, rp);
        });
    } else if (!for_model(c, [&](auto tag) {
            using Mo = typename decltype(tag)::type;
            hipLaunchKernelGGL((k_rollout<Mo>), dim3(grid_for(M)), dim3(kBlock), 0, c->stream, k, g, step_limit, on.dev, ot.dev, M, tr, rp);
        })) return NO_MODEL(c);
    KCHECK();
    bool sync = false;
    TRY(flush_out(c, &on, &sync)); TRY(flush_out(c, &ot, &sync)); TRY(flush_out(c, &os, &sync));
    TRY(flush_out(c, &oa, &sync)); TRY(flush_out(c, &orw, &sync)); TRY(flush_out(c, &otm, &sync));
    if (sync) HIP_TRY(hipStreamSynchronize(c->stream));
    return RSRL_HIP_OK;
}
int rsrl_hip_rollout_greedy(rsrl_hip_ctx* c, int64_t step_limit, uint32_t* n_states_out, float* total_reward_out) {
    CHECK_CTX(c);
    return rollout_impl(c, step_limit, c->cfg.n_envs, n_states_out, total_reward_out, nullptr, nullptr, nullptr, nullptr, RolloutPolicy{});
}
int rsrl_hip_rollout_trajectory(rsrl_hip_ctx* c, int64_t step_limit, int64_t M, uint32_t* n_states_out, float* total_reward_out,
                                float* states_out, int32_t* actions_out, float* rewards_out, uint8_t* terminal_out) {
    return rollout_impl(c, step_limit, M, n_states_out, total_reward_out, states_out, actions_out, rewards_out, terminal_out, RolloutPolicy{});
}
int rsrl_hip_rollout_policy(rsrl_hip_ctx* c, int policy, double epsilon, double tau, int64_t step_limit, int64_t M, uint32_t* n_states_out,
                            float* total_reward_out, float* states_out, int32_t* actions_out, float* rewards_out, uint8_t* terminal_out) {
    CHECK_CTX(c);
    if (policy < 0 || policy > RSRL_RANDOM) return fail(RSRL_HIP_EINVAL, "unknown policy %d", policy);
    if (policy == RSRL_EPSILON_GREEDY && !(epsilon >= 0.0 && epsilon <= 1.0)) return fail(RSRL_HIP_EINVAL, "epsilon must be in [0,1]");      // gen_bool panics otherwise
    if (policy == RSRL_SOFTMAX && std::fabs(tau) < 1e-7) return fail(RSRL_HIP_EINVAL, "Tau parameter in Softmax must be non-zero.");     // softmax.rs:63-66
    RolloutPolicy rp{};
    rp.sample = 1; rp.pp.kind = policy;
    const double v = epsilon * 16777216.0;
    rp.pp.eps_thr = v <= 0.0 ? 0u : (v >= 16777216.0 ? 16777216u : (uint32_t)v);
    rp.pp.eps = (float)epsilon; rp.pp.tau = (float)tau;
    rp.call = c->rollout_calls;
    const int rc = rollout_impl(c, step_limit, M, n_states_out, total_reward_out, states_out, actions_out, rewards_out, terminal_out, rp);
    if (rc == RSRL_HIP_OK) c->rollout_calls++;
    return rc;
}

int rsrl_hip_checksum(rsrl_hip_ctx* c, uint64_t out[2]) {
    CHECK_CTX(c); FLUSH(c);
    if (!out) return fail(RSRL_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    TRY(scratch_reserve(c, 7, 2 * sizeof(unsigned long long)));
    unsigned long long* d = (unsigned long long*)c->scratch[7].p;
    HIP_TRY(hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), c->stream));
    auto run = [&](const void* p, size_t bytes, size_t off, int slot) {
        const size_t n = bytes / 4;
        if (!p || n == 0) return;
        const unsigned g = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(k_checksum, dim3(g), dim3(256), 0, c->stream, (const uint32_t*)p, n, off, d + slot);
    };
    const size_t N = (size_t)c->cfg.n_envs;
    if (c->w_ls != 1) hipLaunchKernelGGL(k_checksum_lm, dim3(4096), dim3(256), 0, c->stream, (const uint32_t*)c->W, (int64_t)N, c->A * c->F, d);
    else run(c->W, c->w_bytes, 0, 0);
    run(c->Z, c->Z ? c->z_bytes : 0, (size_t)1 << 40, 0);
    run(c->state, sizeof(float) * c->D * N, 0, 1);
    run(c->action, sizeof(int32_t) * N, (size_t)1 << 36, 1);
    run(c->ep_step, sizeof(uint32_t) * N, (size_t)1 << 37, 1);
    KCHECK();
    unsigned long long h[2];
    HIP_TRY(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out[0] = h[0]; out[1] = h[1];
    return peer_check(c);
}

int rsrl_hip_fx_saturations(rsrl_hip_ctx* c, uint64_t* count_out) {
    CHECK_CTX(c); FLUSH(c);
    if (!count_out) return fail(RSRL_HIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned int n = 0;
    HIP_TRY(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fx_saturations), sizeof(n), 0, hipMemcpyDeviceToHost));
    *count_out = n;
    return RSRL_HIP_OK;
}

int rsrl_hip_comm_unique_id(uint8_t* id_bytes) {
    if (!id_bytes) return fail(RSRL_HIP_EINVAL, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the ABI");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_bytes, &id, sizeof(id));
    return RSRL_HIP_OK;
}
int rsrl_hip_comm_init(rsrl_hip_ctx* c, const uint8_t* id_bytes, int world_size, int rank) {
    CHECK_CTX(c); FLUSH(c);
    if (!id_bytes || world_size < 1 || rank < 0 || rank >= world_size) return fail(RSRL_HIP_EINVAL, "bad communicator arguments");
    if (c->comm) return fail(RSRL_HIP_ESTATE, "communicator already initialised");
    if (c->cfg.weight_mode != RSRL_W_SHARED) return fail(RSRL_HIP_ESTATE, "per-env weights need no collective: shard by env_offset instead");
    HIP_TRY(hipSetDevice(c->cfg.device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    if (c->multi) return fail(RSRL_HIP_ESTATE, "an exchange is already attached");
    if (c->cfg.exchange == RSRL_EXCHANGE_PEER) return fail(RSRL_HIP_ESTATE, "this ctx was configured for the peer exchange: use rsrl_hip_peer_export / _connect");
    // (AUTO: attaching a communicator decides -- but only once it IS attached: a failed attach leaves the ctx as configured, so that a host
    // can still fall back to the other exchange)
    {
        const ncclResult_t nr = ncclCommInitRank(&c->comm, world_size, id, rank);
        if (nr != ncclSuccess) { c->comm = nullptr; return fail(RSRL_HIP_ERCCL, "ncclCommInitRank: %s", ncclGetErrorString(nr)); }
    }
    // warm-up: RCCL sets its connections up lazily, at the first collective -- which must not be the one inside the step graph's
    // stream capture.  dW is zero between operations, so all-reducing it leaves it zero; every rank makes this call (comm_init is
    // collective by nature).
    {
        const ncclResult_t nr = ncclAllReduce(c->dW, c->dW, c->dw_elems, ncclFloat, ncclSum, c->comm, c->stream);
        const hipError_t he = nr == ncclSuccess ? hipStreamSynchronize(c->stream) : hipSuccess;
        if (nr != ncclSuccess || he != hipSuccess) {
            (void)ncclCommAbort(c->comm); c->comm = nullptr; (void)hipGetLastError();
            return nr != ncclSuccess ? fail(RSRL_HIP_ERCCL, "warm-up all-reduce: %s", ncclGetErrorString(nr))
                                     : fail(RSRL_HIP_EHIP, "warm-up all-reduce: %s", hipGetErrorString(he));
        }
    }
    c->cfg.exchange = RSRL_EXCHANGE_RCCL;
    c->world_size = world_size; c->rank = rank; c->multi = true;
    return RSRL_HIP_OK;
}

int rsrl_hip_comm_info(rsrl_hip_ctx* c, int* world_size, int* rank, int* exchange) {
    CHECK_CTX(c);
    int w = 1, r = 0;
    if (c->comm) { NCCL_TRY(ncclCommCount(c->comm, &w)); NCCL_TRY(ncclCommUserRank(c->comm, &r)); }       // what RCCL itself reports
    else if (c->multi) { w = c->world_size; r = c->rank; }
    if (world_size) *world_size = w;
    if (rank) *rank = r;
    if (exchange) *exchange = !c->multi ? -1 : c->cfg.exchange;
    return RSRL_HIP_OK;
}

// ---- RSRL_EXCHANGE_PEER set-up: export this rank's receive buffer, connect to everybody's -----------------------------
// what a rank tells the others about itself: where its receive buffer is -- and what rsrl_hip_peer_connect needs to decide, the same way
// on every rank, whether the group runs the persistent kernel: the rank's grid (rows), what its device admits (budgets), which
// physical device that is (ranks of one node may share one), and whether it could run the kernel at all (flags bit 0)
struct PeerBlob { uint32_t magic; int32_t pid; uint64_t ptr; uint64_t bytes; int32_t world; uint32_t sh_rows; hipIpcMemHandle_t h;
                  uint64_t dev_id; uint32_t budget_shared; uint32_t flags; };
// identity of the physical device behind a ctx, the same in every process of the node (ordinals are not: HIP_VISIBLE_DEVICES)
static uint64_t device_identity(int device) {
    int dom = 0, bus = 0, dev = 0;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess) { (void)hipGetLastError(); dom = 0; }
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess) { (void)hipGetLastError(); bus = device; }
    if (hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return ((uint64_t)(uint32_t)dom << 32) | ((uint64_t)(uint32_t)(bus & 0xffff) << 16) | (uint64_t)(uint32_t)(dev & 0xffff) | (1ull << 63);
}
static_assert(sizeof(PeerBlob) <= RSRL_HIP_PEER_HANDLE_BYTES, "peer handle blob must fit the ABI slot");
int rsrl_hip_peer_export(rsrl_hip_ctx* c, int world_size, uint8_t* handle_out) {
    CHECK_CTX(c); FLUSH(c);
    if (!handle_out || world_size < 1 || world_size > 64) return fail(RSRL_HIP_EINVAL, "bad peer arguments");
    if (c->cfg.weight_mode != RSRL_W_SHARED) return fail(RSRL_HIP_ESTATE, "per-env weights need no exchange: shard by env_offset instead");
    if (c->cfg.exchange == RSRL_EXCHANGE_RCCL) return fail(RSRL_HIP_ESTATE, "this ctx was configured for the RCCL exchange: use rsrl_hip_comm_init");
    if (c->multi || c->peer_recv) return fail(RSRL_HIP_ESTATE, "an exchange is already attached");
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->peer_old_bytes = sizeof(uint2) * 2 * (size_t)world_size * c->dw_elems;
    // second region: the hop-2 buffer of the persistent kernel, [2 (parity)][world][A*F rounded up to even] granules
    c->peer_recv_bytes = c->peer_old_bytes + sizeof(unsigned long long) * 2 * (size_t)world_size * (((size_t)c->dw_elems + 1) / 2 * 2);
    // fine-grained (uncached across agents) memory, as RCCL uses for its own flag/buffer exchange; RSRL_PEER_COARSE=1 falls
    // back to a plain allocation (same-device peers only need the system-scope accesses the kernels already use)
    hipError_t e = getenv("RSRL_PEER_COARSE") ? hipErrorNotSupported
                                             : hipExtMallocWithFlags((void**)&c->peer_recv, c->peer_recv_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); c->peer_recv = nullptr; e = hipMalloc((void**)&c->peer_recv, c->peer_recv_bytes); }
    PeerBlob b; memset(&b, 0, sizeof(b));
    if (e == hipSuccess) e = hipMemsetAsync(c->peer_recv, 0, c->peer_recv_bytes, c->stream);      // tag 0 never matches a batch-step (tags start at 1)
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&b.h, c->peer_recv);
    if (e != hipSuccess) {
        // (AUTO: exporting a receive buffer decides -- but only once it HAS been exported: a failed export leaves the ctx as configured and
        // without a receive buffer, so that a host can still attach the RCCL exchange)
        (void)hipGetLastError();
        if (c->peer_recv) { (void)hipFree(c->peer_recv); c->peer_recv = nullptr; }
        c->peer_recv_bytes = 0; c->peer_old_bytes = 0;
        return fail(e == hipErrorOutOfMemory ? RSRL_HIP_ENOMEM : RSRL_HIP_EHIP, "exporting the receive buffer: %s", hipGetErrorString(e));
    }
    c->cfg.exchange = RSRL_EXCHANGE_PEER;
    c->peer_world = world_size;
    b.magic = 0x52504552u; b.pid = (int32_t)getpid(); b.ptr = (uint64_t)(uintptr_t)c->peer_recv; b.bytes = c->peer_recv_bytes; b.world = world_size;
    b.sh_rows = c->sh_rows; b.dev_id = device_identity(c->cfg.device);
    b.budget_shared = persist_budget_shared(c);
    b.flags = persist_capable(c) ? 1u : 0u;                            // (RSRL_NO_PERSIST in this rank's environment included: it travels to the others)
    memset(handle_out, 0, RSRL_HIP_PEER_HANDLE_BYTES);
    memcpy(handle_out, &b, sizeof(b));
    return RSRL_HIP_OK;
}
/* identity of the physical device behind ordinal `device` -- PCI domain : bus : device, the same number in every process of the node whatever
 * HIP_VISIBLE_DEVICES each of them runs under (ordinals are not) */
int rsrl_hip_device_identity(int device, uint64_t* identity_out) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (!identity_out) return fail(RSRL_HIP_EINVAL, "null argument");
    if (device < 0 || device >= n) return fail(RSRL_HIP_EINVAL, "device out of range (%d devices)", n);
    *identity_out = device_identity(device);
    return RSRL_HIP_OK;
}
int rsrl_hip_can_access_peer(int device, int peer_device) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n || peer_device < 0 || peer_device >= n) return fail(RSRL_HIP_EINVAL, "device out of range (%d devices)", n);
    if (device == peer_device) return 1;
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, device, peer_device));
    return can ? 1 : 0;
}
int rsrl_hip_peer_connect(rsrl_hip_ctx* c, const uint8_t* handles, int world_size, int rank) {
    CHECK_CTX(c); FLUSH(c);
    if (!handles || world_size < 1 || rank < 0 || rank >= world_size) return fail(RSRL_HIP_EINVAL, "bad peer arguments");
    if (!c->peer_recv || c->peer_world != world_size) return fail(RSRL_HIP_ESTATE, "call rsrl_hip_peer_export(world_size) first");
    if (c->multi) return fail(RSRL_HIP_ESTATE, "an exchange is already attached");
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->peer_ptrs.assign((size_t)world_size, nullptr); c->peer_opened.assign((size_t)world_size, 0);
    for (int r = 0; r < world_size; ++r) {
        PeerBlob b; memcpy(&b, handles + (size_t)r * RSRL_HIP_PEER_HANDLE_BYTES, sizeof(b));
        if (b.magic != 0x52504552u || b.world != world_size || b.bytes != c->peer_recv_bytes)
            return fail(RSRL_HIP_EINVAL, "peer handle %d does not describe a matching receive buffer", r);
        if (r == rank) {
            if ((uint64_t)(uintptr_t)c->peer_recv != b.ptr || b.pid != (int32_t)getpid()) return fail(RSRL_HIP_EINVAL, "handle %d is not this ctx's own export", r);
            c->peer_ptrs[r] = c->peer_recv;
        } else if (b.pid == (int32_t)getpid()) {
            // a ctx of this very process (several ranks driven by one host process): its pointer is valid here -- once this
            // ctx's device may access the memory of the device it lives on
            c->peer_ptrs[r] = (void*)(uintptr_t)b.ptr;
            hipPointerAttribute_t attr;
            HIP_TRY(hipPointerGetAttributes(&attr, c->peer_ptrs[r]));
            if (attr.device != c->cfg.device) {
                int can = 0;
                HIP_TRY(hipDeviceCanAccessPeer(&can, c->cfg.device, attr.device));
                if (!can) return fail(RSRL_HIP_EINVAL, "device %d cannot access the memory of device %d (peer rank %d): use the RCCL exchange", c->cfg.device, attr.device, r);
                const hipError_t pe = hipDeviceEnablePeerAccess(attr.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    return fail(RSRL_HIP_EHIP, "hipDeviceEnablePeerAccess(%d) from device %d: %s", attr.device, c->cfg.device, hipGetErrorString(pe));
                (void)hipGetLastError();
            }
        } else {
            HIP_TRY(hipIpcOpenMemHandle(&c->peer_ptrs[r], b.h, hipIpcMemLazyEnablePeerAccess));
            c->peer_opened[r] = 1;
        }
    }
    HIP_TRY(hipMalloc((void**)&c->d_peer_ptrs, sizeof(void*) * (size_t)world_size));
    HIP_TRY(hipMemcpy(c->d_peer_ptrs, c->peer_ptrs.data(), sizeof(void*) * (size_t)world_size, hipMemcpyHostToDevice));
    {   // the persistent kernel's hop-2 buffers: the second region of every rank's receive buffer
        std::vector<unsigned long long*> bp((size_t)world_size);
        for (int r = 0; r < world_size; ++r) bp[(size_t)r] = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->peer_ptrs[(size_t)r]) + c->peer_old_bytes);
        if (c->px_B && c->px_B_owned) { HIP_TRY(hipFree(c->px_B)); }
        c->px_B = bp[(size_t)rank]; c->px_B_owned = false;
        if (c->d_px_Bptrs) { HIP_TRY(hipFree(c->d_px_Bptrs)); c->d_px_Bptrs = nullptr; }
        HIP_TRY(hipMalloc((void**)&c->d_px_Bptrs, sizeof(void*) * (size_t)world_size));
        HIP_TRY(hipMemcpy(c->d_px_Bptrs, bp.data(), sizeof(void*) * (size_t)world_size, hipMemcpyHostToDevice));
        c->px_seq = 0;                                      // a fresh (cleared) hop-2 buffer: the sequence restarts, on every rank alike
        if (c->px_A) { HIP_TRY(hipFree(c->px_A)); c->px_A = nullptr; }
    }
    {   // (2) the persistent kernel or the per-step kernels: ONE decision for the whole group, computed by every rank from the same
        // handles.  Persistent iff every rank could run it alone AND, on every device that hosts several ranks, the sum of their grids
        // fits the smallest budget any of them reported for it.
        std::vector<PeerBlob> bl((size_t)world_size);
        for (int r = 0; r < world_size; ++r) memcpy(&bl[(size_t)r], handles + (size_t)r * RSRL_HIP_PEER_HANDLE_BYTES, sizeof(PeerBlob));
        bool all = true;
        uint64_t token = 1469598103934665603ull;
        for (int r = 0; r < world_size; ++r) {
            const PeerBlob& b = bl[(size_t)r];
            if (!(b.flags & 1u)) all = false;
            uint64_t rows = 0, budget = ~0ull; int here = 0;
            for (int q = 0; q < world_size; ++q)
                if (bl[(size_t)q].dev_id == b.dev_id) { rows += bl[(size_t)q].sh_rows; if (bl[(size_t)q].budget_shared < budget) budget = bl[(size_t)q].budget_shared; ++here; }
            if (here > 1 && rows > budget) all = false;
            for (uint64_t v : {(uint64_t)(uint32_t)b.pid, b.ptr}) { token ^= v; token *= 1099511628211ull; }
        }
        c->group_persist = all;
        c->group_token = token | 1ull;
        c->coop_allowed = true;
        for (int r = 0; r < world_size; ++r)
            if (r != rank && bl[(size_t)r].pid == bl[(size_t)rank].pid && bl[(size_t)r].dev_id == bl[(size_t)rank].dev_id) c->coop_allowed = false;
        c->coop_validated = false; c->persist_refused = false;
        c->peer_share = 0;
        for (int r = 0; r < world_size; ++r) if (bl[(size_t)r].dev_id == bl[(size_t)rank].dev_id) c->peer_share += 1;
    }
    c->world_size = world_size; c->rank = rank; c->multi = true;
    return RSRL_HIP_OK;
}

// ---- single-process group: every rank is a ctx of THIS process (SURVEY 8b last row; the reference's owner graph is single-threaded,
// rsrl/src/core.rs:13-15, so a Rust host cannot run one blocking ncclCommInitRank per ctx).  One call attaches an exchange to all
// of them, rank = index:
//   RSRL_EXCHANGE_PEER  export + connect of every ctx (same-process pointers; peer access enabled between the devices)
//   RSRL_EXCHANGE_RCCL  ncclCommInitAll over the ctxs' devices (distinct devices, RCCL's rule), then one grouped warm-up
//                       all-reduce so that connection set-up, which needs every rank, does not happen inside the first train()
// Afterwards a single host thread drives the ranks by calling rsrl_hip_train on each ctx in turn: the calls only enqueue.
int rsrl_hip_group_create(rsrl_hip_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1 || n > 64) return fail(RSRL_HIP_EINVAL, "bad group arguments");
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        if (!c) return fail(RSRL_HIP_EINVAL, "null ctx in the group");
        for (int j = 0; j < i; ++j) if (ctxs[j] == c) return fail(RSRL_HIP_EINVAL, "ctx %d appears twice in the group", i);
        FLUSH(c);
        if (c->cfg.weight_mode != RSRL_W_SHARED) return fail(RSRL_HIP_ESTATE, "per-env weights need no exchange: shard by env_offset instead");
        if (c->multi || c->comm || c->peer_recv) return fail(RSRL_HIP_ESTATE, "ctx %d already has an exchange attached", i);
        if (c->cfg.exchange != ctxs[0]->cfg.exchange || c->dw_elems != ctxs[0]->dw_elems || c->cfg.basis != ctxs[0]->cfg.basis)
            return fail(RSRL_HIP_EINVAL, "the ctxs of a group must share the approximator's shape and the exchange kind");
    }
    if (ctxs[0]->cfg.exchange == RSRL_EXCHANGE_AUTO) {
        // PEER whenever every device of the group reaches every other one's memory (one hop, exact integer sums, the persistent kernel);
        // RCCL, the any-topology fallback, otherwise -- or when ranks share a device but RCCL was not asked for explicitly (it needs one
        // device per rank)
        bool peer = true;
        for (int i = 0; i < n && peer; ++i)
            for (int j = 0; j < n && peer; ++j) {
                const int a = ctxs[i]->cfg.device, b = ctxs[j]->cfg.device;
                if (a == b) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess) { (void)hipGetLastError(); can = 0; }
                if (!can) peer = false;
            }
        for (int i = 0; i < n; ++i) ctxs[i]->cfg.exchange = peer ? RSRL_EXCHANGE_PEER : RSRL_EXCHANGE_RCCL;
    }
    if (ctxs[0]->cfg.exchange == RSRL_EXCHANGE_PEER) {
        std::vector<uint8_t> handles((size_t)n * RSRL_HIP_PEER_HANDLE_BYTES);
        for (int i = 0; i < n; ++i) TRY(rsrl_hip_peer_export(ctxs[i], n, handles.data() + (size_t)i * RSRL_HIP_PEER_HANDLE_BYTES));
        for (int i = 0; i < n; ++i) TRY(rsrl_hip_peer_connect(ctxs[i], handles.data(), n, i));
        return RSRL_HIP_OK;
    }
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        devs[(size_t)i] = ctxs[i]->cfg.device;
        for (int j = 0; j < i; ++j)
            // (a collectives library that admits ranks sharing a device says so by exporting `rccl_stub_allows_shared_device` -- the test double of
            // tests/stubs/rccl_stub.cpp, LD_PRELOADed, which exercises this path on a one-GPU box; real RCCL has no such symbol and the check stands)
            if (devs[(size_t)j] == devs[(size_t)i] && !dlsym(RTLD_DEFAULT, "rccl_stub_allows_shared_device"))
                return fail(RSRL_HIP_EINVAL, "RCCL needs one device per rank: ctxs %d and %d share device %d (use RSRL_EXCHANGE_PEER)", j, i, devs[(size_t)i]);
    }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    NCCL_TRY(ncclCommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) { ctxs[i]->comm = comms[(size_t)i]; ctxs[i]->world_size = n; ctxs[i]->rank = i; ctxs[i]->multi = true; ctxs[i]->st_rccl_group = n > 1; }
    // warm-up: dW is zero between operations, so the grouped all-reduce leaves it zero
    NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        HIP_TRY(hipSetDevice(c->cfg.device));
        NCCL_TRY(ncclAllReduce(c->dW, c->dW, c->dw_elems, ncclFloat, ncclSum, c->comm, c->stream));
    }
    NCCL_TRY(ncclGroupEnd());
    for (int i = 0; i < n; ++i) { HIP_TRY(hipSetDevice(ctxs[i]->cfg.device)); HIP_TRY(hipStreamSynchronize(ctxs[i]->stream)); }
    return RSRL_HIP_OK;
}

// One thread stepping every rank of a group it created with rsrl_hip_group_create.  What makes this an entry point of its own:
//  * RCCL: a thread that drives several communicators must issue each collective for ALL of them inside one ncclGroupStart / End;
//    un-grouped, an all-reduce of rank 0 may wait for a rank the same thread has not reached yet, and a train call enqueues hundreds
//    of them.  So the batch-steps advance in lock-step here -- every rank's step kernels, then every rank's all-reduce in ONE group,
//    then what follows the exchange -- and rsrl_hip_train / rsrl_hip_handle refuse such a ctx (RSRL_HIP_ESTATE).
//  * PEER: a rank's exchange kernels wait (bounded) for its peers' kernels, which this same thread has yet to enqueue: the ranks
//    are therefore fed in turns of at most 32 batch-steps, so that no rank's launch queue can fill up in front of a peer that has
//    nothing enqueued (the persistent kernel is one launch per rank and call: no turns needed).
// Results are those of rsrl_hip_train on every rank from a thread of its own, bit for bit.
int rsrl_hip_group_train(rsrl_hip_ctx* const* ctxs, int n, int64_t n_steps) {
    if (!ctxs || n < 1 || n > 64) return fail(RSRL_HIP_EINVAL, "bad group arguments");
    if (n_steps < 0) return fail(RSRL_HIP_EINVAL, "n_steps < 0");
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        if (!c) return fail(RSRL_HIP_EINVAL, "null ctx in the group");
        if (!c->multi || c->world_size != n || c->rank != i || c->cfg.exchange != ctxs[0]->cfg.exchange ||
            (c->cfg.exchange == RSRL_EXCHANGE_PEER && c->group_token != ctxs[0]->group_token))
            return fail(RSRL_HIP_ESTATE, "ctxs[0..%d) must be exactly the ranks of one group made by rsrl_hip_group_create, in rank order", n);
        FLUSH(c);
    }
    if (n_steps == 0) return RSRL_HIP_OK;
    if (ctxs[0]->cfg.exchange == RSRL_EXCHANGE_PEER) {
        const int64_t turn = persist_ok(ctxs[0]) ? n_steps : kStepsPerGraph;
        for (int64_t done = 0; done < n_steps; done += turn)
            for (int i = 0; i < n; ++i) TRY(train_now(ctxs[i], n_steps - done < turn ? n_steps - done : turn, nullptr));
        return RSRL_HIP_OK;
    }
    // RCCL: lock-step, the all-reduces of a batch-step grouped
    std::vector<Common> ks((size_t)n);
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        HIP_TRY(hipSetDevice(c->cfg.device));
        if (c->sh_tab && c->t != c->sh_tab_t) HIP_TRY(hipMemsetAsync(c->sh_tab, 0, sizeof(long long) * 3 * kTabRep * c->dw_elems, c->stream));
    }
    for (int64_t j = 0; j < n_steps; ++j) {
        for (int i = 0; i < n; ++i) {
            rsrl_hip_ctx* c = ctxs[i];
            HIP_TRY(hipSetDevice(c->cfg.device));
            ks[(size_t)i] = make_common(c);
            TRY(enqueue_shared_step(c, ks[(size_t)i], make_geom(c), nullptr, j == 0 ? 0 : 1, c->t, nullptr, 1));
        }
        NCCL_TRY(ncclGroupStart());
        for (int i = 0; i < n; ++i) {
            rsrl_hip_ctx* c = ctxs[i];
            HIP_TRY(hipSetDevice(c->cfg.device));
            if (c->sh_tab) TRY(exchange_table(c, c->t));                  // dense basis: the batch-step's fixed-point table
            else NCCL_TRY(ncclAllReduce(c->dW, c->dW, c->dw_elems, ncclFloat, ncclSum, c->comm, c->stream));      // tile coding: the float delta
        }
        NCCL_TRY(ncclGroupEnd());
        for (int i = 0; i < n; ++i) {
            rsrl_hip_ctx* c = ctxs[i];
            HIP_TRY(hipSetDevice(c->cfg.device));
            TRY(enqueue_shared_step(c, ks[(size_t)i], make_geom(c), nullptr, j == 0 ? 0 : 1, c->t, nullptr, 2));
            c->t += 1;
            c->kernel_name = shared_kernel_name(c);
        }
    }
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        HIP_TRY(hipSetDevice(c->cfg.device));
        TRY(enqueue_shared_c(c, make_common(c), make_geom(c), c->t - 1));
        c->sh_tab_t = c->t;
    }
    return RSRL_HIP_OK;
}

int rsrl_hip_timing_enable(rsrl_hip_ctx* c, int enable) {
    CHECK_CTX(c); FLUSH(c);
    c->timing = enable != 0;
    c->events_used = 0;
    return RSRL_HIP_OK;
}
int rsrl_hip_timing_read(rsrl_hip_ctx* c, double* ms_total, uint64_t* launches, const char** kernel_name) {
    CHECK_CTX(c); FLUSH(c);
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (size_t i = 0; i < c->events_used; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->events[i].first, c->events[i].second));
        tot += ms;
    }
    if (ms_total) *ms_total = tot;
    if (launches) { uint64_t n = 0; for (size_t i = 0; i < c->events_used; ++i) n += c->event_launches[i]; *launches = n; }
    if (kernel_name) *kernel_name = c->kernel_name;
    return RSRL_HIP_OK;
}

}  // extern "C"
