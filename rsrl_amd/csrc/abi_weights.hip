// abi_weights.hip -- Parameterised (params/mod.rs:116-134): weights, traces, fa_td's weights in the reference's (F, A) order; the checkpoint
// format (include/rsrl_hip.h); checksums.
#include "ctx.hpp"

RSRL_API_BEGIN

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
int traces_rw(rsrl_hip_ctx* c, int64_t env_index, float* out, const float* in) {
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
        hipLaunchKernelGGL(k_sparse_trace_get, dim3(kSparseCap / 256), dim3(256), 0, c->stream, SparseTrace{c->sp_keys, c->sp_vals, c->sp_len}, c->cfg.n_tilings, n / c->cfg.n_tilings, env_index, oz.dev);
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
        const uint32_t slice = (uint32_t)c->F * (uint32_t)c->Aw / (uint32_t)T;                 // entries of one tiling's slice: the device's keys are relative to it
        std::vector<uint32_t> lens((size_t)N * T), tot((size_t)N);
        std::vector<uint16_t> keys((size_t)(kSparseChunk * kSparseCap));
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
            e = hipMemcpyAsync(keys.data(), c->sp_keys + i0 * kSparseCap, 2 * (size_t)(n * kSparseCap), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(vals.data(), c->sp_vals + i0 * kSparseCap, 4 * (size_t)(n * kSparseCap), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { rc = fail(RSRL_HIP_EHIP, "reading the sparse traces: %s", hipGetErrorString(e)); break; }
            for (int64_t i = 0; rc == RSRL_HIP_OK && i < n; ++i) {
                size_t l = 0;
                for (int t = 0; t < T; ++t)
                    for (uint32_t j = 0; j < lens[(size_t)(i0 + i) * T + t]; ++j, ++l) {
                        kk[l] = (uint32_t)t * slice + (uint32_t)keys[(size_t)(i * kSparseCap + t * cap) + j];      // (the file holds FULL keys: tile index * A + action)
                        vv[l] = vals[(size_t)(i * kSparseCap + t * cap) + j];
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
    uint16_t* spk_new = nullptr; float* spv_new = nullptr;              // sparse traces: shadow lists, switched in at the end like W
    if (rc == RSRL_HIP_OK && h.aux_kind == 4) {
        const int64_t N = c->cfg.n_envs;
        hipError_t e2 = hipMalloc((void**)&spk_new, 2 * (size_t)kSparseCap * (size_t)N);
        if (e2 == hipSuccess) e2 = hipMalloc((void**)&spv_new, 4 * (size_t)kSparseCap * (size_t)N);
        if (e2 != hipSuccess) rc = fail(e2 == hipErrorOutOfMemory ? RSRL_HIP_ENOMEM : RSRL_HIP_EHIP, "staging buffers for the sparse traces: %s", hipGetErrorString(e2));
        std::vector<uint16_t> keys((size_t)(kSparseChunk * kSparseCap));
        std::vector<uint32_t> kk((size_t)kSparseCap);
        std::vector<float> vals((size_t)(kSparseChunk * kSparseCap)), vv((size_t)kSparseCap);
        if (rc == RSRL_HIP_OK && fseek(f, sp_prefix + 4 * (long)N, SEEK_CUR) != 0) rc = fail(RSRL_HIP_EINVAL, "%s: read error", path);      // (owner and lengths: read above)
        const int T = c->cfg.n_tilings, cap = kSparseCap / T;
        const uint32_t n_keys = (uint32_t)c->F * (uint32_t)c->Aw, slice = n_keys / (uint32_t)T;
        sp_len_t.assign((size_t)N * T, 0u);
        for (int64_t i0 = 0; rc == RSRL_HIP_OK && i0 < N; i0 += kSparseChunk) {
            const int64_t n = std::min<int64_t>(kSparseChunk, N - i0);
            std::fill(keys.begin(), keys.end(), (uint16_t)0); std::fill(vals.begin(), vals.end(), 0.0f);
            for (int64_t i = 0; rc == RSRL_HIP_OK && i < n; ++i) {
                const size_t l = sp_len_in[(size_t)(i0 + i)];
                if (fread(kk.data(), 4, l, f) != l || fread(vv.data(), 4, l, f) != l) rc = fail(RSRL_HIP_EINVAL, "%s: read error", path);
                for (size_t k = 0; rc == RSRL_HIP_OK && k < l; ++k) {         // every entry into the sub-list of its key's tiling
                    if (kk[k] >= n_keys) { rc = fail(RSRL_HIP_EINVAL, "%s: corrupt sparse trace of learner %lld (key out of range)", path, (long long)(i0 + i)); break; }
                    const uint32_t t = kk[k] / slice; uint32_t& lt = sp_len_t[(size_t)(i0 + i) * T + t];
                    if (lt >= (uint32_t)cap) { rc = fail(RSRL_HIP_EINVAL, "%s: learner %lld's sparse trace holds more than %d entries of tiling %u (this library keeps "
                                                                            "%d entries per learner as %d per tiling)", path, (long long)(i0 + i), cap, t, kSparseCap, cap); break; }
                    keys[(size_t)(i * kSparseCap + (int64_t)t * cap) + lt] = (uint16_t)(kk[k] - t * slice);      // (relative to the tiling's slice: < slice <= 65 536)
                    vals[(size_t)(i * kSparseCap + (int64_t)t * cap) + lt] = vv[k];
                    lt += 1;
                }
            }
            if (rc != RSRL_HIP_OK) break;
            e2 = hipMemcpyAsync(spk_new + i0 * kSparseCap, keys.data(), 2 * (size_t)(n * kSparseCap), hipMemcpyHostToDevice, c->stream);
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
    unsigned int n[4] = {0, 0, 0, 0};
    if (fx_saturations_train(&n[0]) || fx_saturations_trait(&n[1]) || fx_saturations_util(&n[2]) || fx_saturations_launch(&n[3])) { (void)hipGetLastError(); return fail(RSRL_HIP_EHIP, "reading the saturation counters"); }
    *count_out = (uint64_t)n[0] + n[1] + n[2] + n[3];
    return RSRL_HIP_OK;
}
RSRL_API_END
