// abi_trait.hip -- the trait-granular entry points: Domain::transition, Function / Enumerable, Policy, Handler::handle, and the deferred
// trait loop (kernels_trait.hpp) that runs a batch-step's four calls as one launch.
#include "ctx.hpp"

RSRL_DEFINE_FX_READER(fx_saturations_trait)

// ---- the trait-granular loop (kernels_trait.hpp) --------------------------------------------------------------------------------------
// the hand-over cache is keyed by the state an entry belongs to; whenever the weights changed behind it (train, set / load weights) the keys are
// emptied (NaN: never equal to a state) before the next kernel that looks at them
int trait_cache_ready(rsrl_hip_ctx* c) {
    if (c->tq_valid) return RSRL_HIP_OK;
    HIP_TRY(hipMemsetAsync(c->tq_key, 0xFF, sizeof(float) * (size_t)c->D * (size_t)c->cfg.n_envs, c->stream));
    c->tq_valid = true;
    return RSRL_HIP_OK;
}
int launch_domain_step(rsrl_hip_ctx* c, const Common& k, const int32_t* d_act, float* from, float* next, float* rew, uint8_t* term) {
    const dim3 g(grid_for(c->cfg.n_envs)), b(kBlock);
    switch (c->cfg.domain) {
    case 0: hipLaunchKernelGGL(k_domain_step<0>, g, b, 0, c->stream, k, d_act, from, next, rew, term); break;
    case 1: hipLaunchKernelGGL(k_domain_step<1>, g, b, 0, c->stream, k, d_act, from, next, rew, term); break;
    default: hipLaunchKernelGGL(k_domain_step<2>, g, b, 0, c->stream, k, d_act, from, next, rew, term); break;
    }
    KCHECK();
    return RSRL_HIP_OK;
}
int launch_domain_reset(rsrl_hip_ctx* c, const Common& k, const uint8_t* d_mask) {
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
int launch_trait_handle(rsrl_hip_ctx* c, const Common& k, const float* from, const int32_t* act, const float* rew, const float* to,
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
int trait_flush(rsrl_hip_ctx* c) {
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

int flush_pending(rsrl_hip_ctx* c) {
    if (!c) return RSRL_HIP_OK;
    if (c->tp.stage) TRY(trait_flush(c));
    if (c->pending == 0) return RSRL_HIP_OK;
    const int64_t n = c->pending;
    c->pending = 0;
    return train_now(c, n, nullptr);
}
// fused register-family loop: any split of n batch-steps into launches gives bit-identical results (Q(s,.) is carried between

RSRL_API_BEGIN

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
        if (!for_model(c, [&](auto tag) {
                using Mo = typename decltype(tag)::type;
                if constexpr (Mo::kSparse) {
                    hipLaunchKernelGGL((k_sparse_handle<Mo>), dim3(grid_for(M)), dim3(kBlock), 0, c->stream, ks, g, d_from, d_act, d_rew, d_to, d_term, M, c->t,
                                       c->cfg.algo == RSRL_Q_LAMBDA ? 1 : 0, c->flags, c->sc_keys, c->sc_terms, otd.dev);
                    launch_sparse_trace_scatter(c, M, 0);
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
        launch_wave_agent(c, k, M, c->t, 1, nullptr, d_from, d_act, d_rew, d_to, d_term, M, otd.dev);
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
        launch_wave_agent(c, k, M, c->t, 1, nullptr, d_from, d_act, d_rew, d_to, d_term, M, otd.dev);
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
        launch_wave_agent(c, k, M, c->t, 1, nullptr, d_from, d_act, d_rew, d_to, d_term, M, otd.dev);
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
RSRL_API_END
