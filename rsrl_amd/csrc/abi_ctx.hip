// abi_ctx.hip -- the ctx's life cycle and plain state: create / destroy (the constructors of examples/q_learning.rs:19-32), configuration checks,
// the env state accessors, reset, the timing hooks.
#include "ctx.hpp"

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_last_error = buf;
    // the HIP runtime keeps the last failure of ANY call of this thread until somebody asks for it: asked for here WHEN THE FAILURE REPORTED IS A HIP ONE, so
    // that it is not found again by the next launch check (KCHECK) of a healthy ctx.  A pure argument / state error (EINVAL, ESTATE) leaves it alone: a launch
    // error not yet checked must not be lost behind an unrelated report; rsrl_hip_destroy and the clean-up paths clear it themselves.
    if (code == RSRL_HIP_EHIP || code == RSRL_HIP_ENOMEM) (void)hipGetLastError();
    return code;
}

RSRL_API_BEGIN

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
        if (cfg->weight_mode != RSRL_W_PER_ENV) return fail(RSRL_HIP_EINVAL, "QSigma needs per-learner weights");
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
    // write back only the touched column (RSRL_K1_FEATURE_MAJOR=1 keeps the feature-major layout, for A/B runs).  NOT the default layout: the fused loop runs on it
    // bit for bit, but its strided load / store of W costs 71 against 26 us per 20-step launch and 1.3 % of the coalesced rate (round 6, measured)
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
        HIP_TRY(hipMalloc((void**)&c->sp_keys, sizeof(uint16_t) * (size_t)kSparseCap * (size_t)N));
        HIP_TRY(hipMalloc((void**)&c->sp_vals, sizeof(float) * (size_t)kSparseCap * (size_t)N));
        HIP_TRY(hipMalloc((void**)&c->sp_len, sizeof(uint32_t) * (size_t)cfg->n_tilings * (size_t)N));
        HIP_TRY(hipMemsetAsync(c->sp_len, 0, sizeof(uint32_t) * (size_t)cfg->n_tilings * (size_t)N, c->stream));        // Trace::zeros: empty lists
        // (the lists are written only below their lengths; what lies beyond is never read as an entry, but a checkpoint copies whole rows)
        HIP_TRY(hipMemsetAsync(c->sp_keys, 0, sizeof(uint16_t) * (size_t)kSparseCap * (size_t)N, c->stream));
        HIP_TRY(hipMemsetAsync(c->sp_vals, 0, sizeof(float) * (size_t)kSparseCap * (size_t)N, c->stream));
        c->sp_lds = slice * 8 <= 128 * 1024;
        if (c->sp_lds && slice * 8 > 64 * 1024) c->sp_lds = sparse_trace_scatter_allow_lds(cfg->n_tilings, (int)(slice * 8));      // more dynamic LDS than a kernel gets by default
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

void state_limits(const rsrl_hip_ctx* c, float* lo, float* hi) {
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
bool carries_q(const rsrl_hip_ctx* c) {          // the kernels that read Common::qcache: the register family's one-step loops
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
RSRL_API_END
