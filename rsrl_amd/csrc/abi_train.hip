// abi_train.hip -- the fused driver loop (examples/q_learning.rs:34-59 x N learners x n batch-steps): launch shapes per kernel family, the captured
// step graphs, the persistent shared-W kernel's co-residency gate, launch coalescing, rollouts.
#include "ctx.hpp"

RSRL_DEFINE_FX_READER(fx_saturations_train)

// ---- the fused driver loop -----------------------------------------------------------------
int timing_begin(rsrl_hip_ctx* c) {
    if (!c->timing) return RSRL_HIP_OK;
    if (c->events_used == c->events.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
        c->events.emplace_back(a, b);
    }
    HIP_TRY(hipEventRecord(c->events[c->events_used].first, c->stream));
    return RSRL_HIP_OK;
}
int timing_end(rsrl_hip_ctx* c, uint32_t launches) {
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
int exchange_table(rsrl_hip_ctx* c, uint64_t t) {
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
int enqueue_shared_step(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, DevStats* d_stats, int do_c, uint64_t t, const uint64_t* t_dev, int xpart) {
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
                    hipLaunchKernelGGL((k_shared_ca<M>), grid, block, 0, c->stream, ks, g, t, do_c | (c->cfg.algo == RSRL_Q_LAMBDA ? 2 : 0), dwp, c->flags, d_stats, nrep,
                                       (int64_t)c->dw_elems, t_dev, c->sc_keys, c->sc_terms);
                    static const int per_env = getenv("RSRL_SPARSE_CHUNK") ? atoi(getenv("RSRL_SPARSE_CHUNK")) : 0;      // (A/B; 0: launch_shared.hip's rule)
                    launch_sparse_trace_scatter(c, (int64_t)k.n_envs, per_env > 0 && per_env < 16 ? 16 : per_env);
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
int enqueue_shared_c(rsrl_hip_ctx* c, const Common& k, const BasisGeom& g, uint64_t t_last) {
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
unsigned persist_budget_shared(rsrl_hip_ctx* c) {
    const int nb = persist_blocks_per_cu(c);
    return nb >= 1 ? (unsigned)c->n_cu * (unsigned)(nb > 1 ? nb - 1 : 1) : 0u;
}
// this rank alone: could it run the persistent kernel?  (multi-rank: what goes into the handle; the group decides)
bool persist_capable(rsrl_hip_ctx* c) {
    if (c->cfg.weight_mode != RSRL_W_SHARED || !c->sh_tab) return false;
    if (getenv("RSRL_NO_PERSIST")) return false;
    return c->sh_rows <= persist_budget_single(c);
}
bool persist_ok(rsrl_hip_ctx* c) {
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
int train_now(rsrl_hip_ctx* c, int64_t n_steps, rsrl_hip_stats* stats_out) {
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
            launch_wave_agent(c, k, k.n_envs, c->t, chunk, d_stats, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
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
            launch_wave_agent(c, k, k.n_envs, c->t, chunk, d_stats, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
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
            launch_wave_agent(c, k, k.n_envs, c->t, chunk, d_stats, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
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
RSRL_API_BEGIN

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
RSRL_API_END
