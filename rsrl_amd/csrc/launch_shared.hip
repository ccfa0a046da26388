// launch_shared.hip -- the launches of the kernel templates that both the trait-granular path (abi_trait.hip) and the driver loop (abi_train.hip) use: the order-7
// wave family's memory-sweep agents and the sparse-trace scatter.  One instantiation, one copy of the machine code (ctx.hpp declares); a unit of its own so that a
// change to these templates does not move the code object of the small shared kernels (kernels_util.hip) and with it their profile digests.
#include "ctx.hpp"

RSRL_DEFINE_FX_READER(fx_saturations_launch)

void launch_wave_agent(const rsrl_hip_ctx* c, const Common& k, int64_t items, uint64_t t, int n_steps, DevStats* d_stats, const float* from, const int32_t* act,
                       const float* rew, const float* to, const uint8_t* term, int64_t M, float* td_out) {
    const dim3 grid(wave_grid_for(items)), block(kBlock);
    for_wave(c, [&](auto tag) {
        using T = decltype(tag); using WT = typename T::wt;
        if (is_wave_aux_algo(c->cfg.algo)) {
            hipLaunchKernelGGL((k_wave_aux<T::domain, WT>), grid, block, 0, c->stream, k, make_wave_aux(c), (WT*)c->W, t, n_steps, d_stats, from, act, rew, to, term, M, td_out);
        } else if (c->cfg.algo == RSRL_Q_SIGMA) {
            hipLaunchKernelGGL((k_wave_qsigma<T::domain, WT>), grid, block, 0, c->stream, k, make_qs(c), (WT*)c->W, t, n_steps, d_stats, from, act, rew, to, term, M, td_out);
        } else {
            hipLaunchKernelGGL((k_wave_lambda<T::domain, WT>), grid, block, 0, c->stream, k, make_lambda(c), (WT*)c->W, t, n_steps, d_stats, from, act, rew, to, term, M, td_out);
        }
    });
}

template <int T>
static void sparse_scatter_launch(const rsrl_hip_ctx* c, int64_t n_learners, int per) {
    const int slice = (int)((int64_t)(c->F / c->cfg.n_tilings) * c->A);
    const float step_size = (float)c->cfg.alpha;
    hipLaunchKernelGGL((k_sparse_trace_scatter<T>), dim3((unsigned)((n_learners + per - 1) / per), (unsigned)T), dim3(1024), c->sp_lds ? (size_t)slice * 8 : 0, c->stream,
                       c->sc_keys, c->sc_terms, c->flags, SparseTrace{c->sp_keys, c->sp_vals, c->sp_len}, make_lambda(c), n_learners, (int64_t)c->cfg.n_envs, slice, per,
                       c->dW_rep, c->n_rep, (int64_t)c->dw_elems, FxScale(step_size).inv_lsb, c->sp_lds ? 1 : 0);
}
// per_block <= 0: ONE block per compute unit over the (chunk, tiling) grid -- every block clears and sweeps its 8 S-byte slice and issues one device atomic per touched
// entry, so fewer, larger blocks win until compute units idle (65 536 CartPole learners, 8 tilings: 72.7 / 66.1 / 62.6 / 104.6 us at 512 / 1 024 / 2 048 / 4 096
// learners per block; 262 144: 269 / 244 / 229 at 512 / 2 048 / 8 192; 8 192: 26.1 / 21.3 / 26.8 at 128 / 256 / 512 -- scripts/gpu_sparse_scatter_ab.sh)
void launch_sparse_trace_scatter(const rsrl_hip_ctx* c, int64_t n_learners, int per_block) {
    if (per_block <= 0) {
        const int64_t cus = c->n_cu > 0 ? c->n_cu : 256;
        int64_t per = (n_learners * c->cfg.n_tilings + cus - 1) / cus;
        per = ((per + 63) / 64) * 64;                                        // (a block's sixteen waves carry 64 or 128 learners at a time)
        per_block = (int)(per < 64 ? 64 : per > (1 << 20) ? (1 << 20) : per);
    }
    if (c->cfg.n_tilings == 4) sparse_scatter_launch<4>(c, n_learners, per_block);
    else if (c->cfg.n_tilings == 8) sparse_scatter_launch<8>(c, n_learners, per_block);
    else sparse_scatter_launch<16>(c, n_learners, per_block);
}
bool sparse_trace_scatter_allow_lds(int n_tilings, int bytes) {
    const void* fn = n_tilings == 4 ? reinterpret_cast<const void*>(&k_sparse_trace_scatter<4>)
                   : n_tilings == 8 ? reinterpret_cast<const void*>(&k_sparse_trace_scatter<8>) : reinterpret_cast<const void*>(&k_sparse_trace_scatter<16>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}
