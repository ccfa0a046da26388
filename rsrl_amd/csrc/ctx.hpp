// ctx.hpp -- what the translation units of the C ABI (abi_*.hip; include/rsrl_hip.h) share: the ctx, error reporting, the (basis, domain, order) -> model
// dispatch, host / device pointer staging, and the small kernels more than one unit launches.  Internal: nothing here is exported.
//   abi_ctx.hip      create / destroy, configuration, env state accessors, reset, timing hooks
//   abi_trait.hip    the trait-granular entry points: Domain::transition, Function / Enumerable, Policy, Handler::handle (+ the deferred trait loop)
//   abi_weights.hip  Parameterised (weights, traces, fa_td), checkpoints, checksums
//   abi_train.hip    the fused driver loop: launch shapes, step graphs, the persistent shared-W kernel, rollouts
//   abi_group.hip    multi-rank: RCCL communicators, the peer exchange set-up, single-process groups
//   kernels_util.hip, launch_shared.hip   the small kernels / the kernel-template launches more than one of those units needs (one copy of the machine code)
#pragma once
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
// everything below is internal to the library: only what include/rsrl_hip.h declared (above, default visibility) is exported
#pragma GCC visibility push(hidden)
#define RSRL_API_BEGIN _Pragma("GCC visibility push(default)") extern "C" {
#define RSRL_API_END } _Pragma("GCC visibility pop")
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

// ---- the small kernels more than one unit launches, and the launches of kernel templates two units would otherwise both instantiate: defined ONCE, in
// kernels_util.hip / launch_shared.hip (a kernel's host stub is an ordinary function: another unit launches it through this declaration)
__global__ __launch_bounds__(256) void k_apply_rep(float* __restrict__ W, float* __restrict__ dW, long long* __restrict__ rep, int n_rep, int n, float lsb);
__global__ __launch_bounds__(1024) void k_tile_scatter(const uint16_t* __restrict__ keys, const float* __restrict__ terms, int64_t N, int S, int per_block, long long* __restrict__ dW64, int n_rep, int64_t rep_stride, float inv_lsb);
__global__ __launch_bounds__(256) void k_fx_finalize(long long* __restrict__ fx, float* __restrict__ dW, int n, float lsb);
__global__ void k_clamp_actions(int32_t* __restrict__ a, int64_t n, int A);
__global__ __launch_bounds__(256) void k_peer_push(const float* __restrict__ dW, int n, uint2* const* __restrict__ peers, int world, int rank, uint64_t t, const uint64_t* __restrict__ t_dev, int64_t xdelta);
__global__ __launch_bounds__(256) void k_peer_reduce(float* __restrict__ dW, int n, const uint2* __restrict__ recv, int world, uint64_t t, const uint64_t* __restrict__ t_dev, int64_t xdelta, uint32_t* __restrict__ err, uint64_t timeout);
__global__ __launch_bounds__(kBlock) void k_tab_finalize(const long long* __restrict__ tab, int n, float lr, float* __restrict__ dW, uint64_t t, const uint64_t* __restrict__ t_dev);
__global__ __launch_bounds__(256) void k_tab_exchange_apply(const long long* __restrict__ tab, int n, float lr, uint2* const* __restrict__ peers, const uint2* __restrict__ recv, float* __restrict__ W, int world, int rank, uint64_t t, const uint64_t* __restrict__ t_dev, int64_t xdelta, uint32_t* __restrict__ err, uint64_t timeout);
__global__ void k_fill_f32(float* __restrict__ p, int64_t n, float v);
__global__ void k_set_dyn(DynParams* __restrict__ d, DynParams v);
__global__ void k_set_t(uint64_t* __restrict__ t_dev, uint64_t v);
__global__ void k_advance_t(uint64_t* __restrict__ t_dev, uint64_t d);
__global__ void k_apply_dw(float* __restrict__ W, float* __restrict__ dW, int n);
__global__ void k_weights_get(const float* __restrict__ W, bool tile, int64_t stride, int64_t wi, int F, int A, float* __restrict__ out);
__global__ void k_weights_set(float* __restrict__ W, bool tile, int64_t stride, int64_t wi, int F, int A, const float* __restrict__ in);
__global__ void k_weights_set_all(float* __restrict__ W, bool tile, int64_t N, int64_t stride, int64_t ls, int F, int A, const float* __restrict__ in);
__global__ void k_checksum_lm(const uint32_t* __restrict__ p, int64_t N, int AF, unsigned long long* __restrict__ out);
__global__ void k_checksum(const uint32_t* __restrict__ p, size_t n, size_t index_offset, unsigned long long* __restrict__ out);
struct rsrl_hip_ctx;
// the order-7 wave family's memory-sweep agents (GreedyGQ / TD / TDLambda: k_wave_aux; QSigma: k_wave_qsigma; SARSALambda / QLambda: k_wave_lambda), driver loop
// (from == nullptr: n_steps batch-steps of the ctx's learners) or Handler::handle on M caller-supplied transitions
void launch_wave_agent(const rsrl_hip_ctx* c, const rsrl::Common& k, int64_t items, uint64_t t, int n_steps, rsrl::DevStats* d_stats, const float* from,
                       const int32_t* act, const float* rew, const float* to, const uint8_t* term, int64_t M, float* td_out);
// the trace update + LDS scatter of the sparse-trace lambda agents for learners 0 .. n_learners-1 (kernels_sparse_lambda.hpp)
void launch_sparse_trace_scatter(const rsrl_hip_ctx* c, int64_t n_learners, int per_block);
// dynamic LDS beyond the default for that kernel (a tiling's slice of more than 64 KiB): false when the runtime refuses
bool sparse_trace_scatter_allow_lds(int n_tilings, int bytes);

// ------------------------------------------------------------------------------- errors
extern thread_local std::string g_last_error;      // (abi_ctx.hip)

int fail(int code, const char* fmt, ...);           // records the message for rsrl_hip_last_error() and returns `code` (abi_ctx.hip)
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
    uint16_t* sp_keys = nullptr; float* sp_vals = nullptr; uint32_t* sp_len = nullptr;    // sparse traces: [N][kSparseCap] slice-relative keys (16 bit) and values, lengths [N][n_tilings]
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
extern "C" __attribute__((visibility("hidden"))) void state_limits(const rsrl_hip_ctx* c, float* lo, float* hi);      // (abi_ctx.hip; internal: hidden visibility)
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
static __global__ void k_check_states(const float* __restrict__ s, int64_t n, int D, StateLimits lim, unsigned* __restrict__ bad) {
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

int flush_pending(rsrl_hip_ctx* c);      // launch what the ctx has accepted but not launched yet (train's coalesced batch-steps, deferred trait calls)
int trait_flush(rsrl_hip_ctx* c);        // ... the deferred trait calls alone, one kernel per call
int timing_begin(rsrl_hip_ctx* c);
int timing_end(rsrl_hip_ctx* c, uint32_t launches = 1);
int trait_cache_ready(rsrl_hip_ctx* c);
static inline bool trait_fast(const rsrl_hip_ctx* c) { return c->tq_key != nullptr; }
int launch_trait_handle(rsrl_hip_ctx* c, const Common& k, const float* from, const int32_t* act, const float* rew, const float* to,
                               const uint8_t* term, int64_t M, uint64_t t, float* td);
int launch_domain_step(rsrl_hip_ctx* c, const Common& k, const int32_t* d_act, float* from, float* next, float* rew, uint8_t* term);
int launch_domain_reset(rsrl_hip_ctx* c, const Common& k, const uint8_t* d_mask);
#define FLUSH(c) TRY(flush_pending(c))

// ---- defined in one unit, used by others
// g_fx_saturations (models.hpp) is a translation-unit-local device counter: every unit whose kernels quantise has its own, rsrl_hip_fx_saturations adds them up
#define RSRL_DEFINE_FX_READER(name) \
    int name(unsigned int* n) { return hipMemcpyFromSymbol(n, HIP_SYMBOL(g_fx_saturations), sizeof(*n), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
int fx_saturations_train(unsigned int* n);      // abi_train.hip: k_shared_step / k_shared_ca / k_shared_persist
int fx_saturations_trait(unsigned int* n);      // abi_trait.hip: k_handle on shared weights
int fx_saturations_util(unsigned int* n);       // kernels_util.hip: k_tile_scatter
int fx_saturations_launch(unsigned int* n);     // launch_shared.hip: k_sparse_trace_scatter
int train_now(rsrl_hip_ctx* c, int64_t n_steps, rsrl_hip_stats* stats_out);                                                              // abi_train.hip
int enqueue_shared_step(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, DevStats* d_stats, int do_c, uint64_t t, const uint64_t* t_dev, int xpart = 0);
int enqueue_shared_c(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, uint64_t t_last);
int exchange_table(rsrl_hip_ctx* c, uint64_t t);
bool persist_capable(rsrl_hip_ctx* c);
bool persist_ok(rsrl_hip_ctx* c);
unsigned persist_budget_shared(rsrl_hip_ctx* c);
extern "C" __attribute__((visibility("hidden"))) bool carries_q(const rsrl_hip_ctx* c);                                                                                                   // abi_ctx.hip
extern "C" __attribute__((visibility("hidden"))) int traces_rw(rsrl_hip_ctx* c, int64_t env_index, float* out, const float* in);                                                          // abi_weights.hip
constexpr int kStepsPerGraph = 32;       // batch-steps per captured step graph (abi_train.hip)
#define ST_RCCL_GUARD(c) do { if ((c)->st_rccl_group) return fail(RSRL_HIP_ESTATE, "this ctx is a rank of a single-thread RCCL group: its collectives must be " \
    "issued for all ranks together -- step the group with rsrl_hip_group_train (or use one thread / process per rank, or RSRL_EXCHANGE_PEER)"); } while (0)
