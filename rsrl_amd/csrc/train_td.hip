// prediction (TD / TDLambda) kernels of the register family
#include "launch.hpp"
#include "kernels_td.hpp"
namespace rsrl {

#define RSRL_TD_CASE(DM, OR)                                                                                        \
    if (domain == DM && order == OR) {                                                                              \
        if (lambda) hipLaunchKernelGGL((k_train_td<DM, OR, true>), grid, block, 0, st, k, tp, t, chunk, stats);     \
        else hipLaunchKernelGGL((k_train_td<DM, OR, false>), grid, block, 0, st, k, tp, t, chunk, stats);           \
        return true;                                                                                                \
    }
bool launch_train_td(int domain, int order, bool lambda, dim3 grid, dim3 block, hipStream_t st, const Common& k, const TdParams& tp,
                     uint64_t t, int chunk, DevStats* stats) {
    RSRL_TD_CASE(0, 1) RSRL_TD_CASE(0, 2) RSRL_TD_CASE(0, 3) RSRL_TD_CASE(0, 4) RSRL_TD_CASE(0, 5) RSRL_TD_CASE(1, 1) RSRL_TD_CASE(2, 1)
    return false;
}
#define RSRL_HTD_CASE(DM, OR)                                                                                            \
    if (domain == DM && order == OR) {                                                                                   \
        hipLaunchKernelGGL((k_handle_td<DM, OR>), grid, block, 0, st, k, tp, lambda ? 1 : 0, from, rew, to, termf, Mn, td_out); \
        return true;                                                                                                     \
    }
bool launch_handle_td(int domain, int order, bool lambda, dim3 grid, dim3 block, hipStream_t st, const Common& k, const TdParams& tp,
                      const float* from, const float* rew, const float* to, const uint8_t* termf, int64_t Mn, float* td_out) {
    RSRL_HTD_CASE(0, 1) RSRL_HTD_CASE(0, 2) RSRL_HTD_CASE(0, 3) RSRL_HTD_CASE(0, 4) RSRL_HTD_CASE(0, 5) RSRL_HTD_CASE(1, 1) RSRL_HTD_CASE(2, 1)
    return false;
}
#define RSRL_VEV_CASE(DM, OR)                                                                            \
    if (domain == DM && order == OR) {                                                                   \
        hipLaunchKernelGGL((k_v_evaluate<DM, OR>), grid, block, 0, st, k, states, Mn, out);              \
        return true;                                                                                     \
    }
bool launch_v_evaluate(int domain, int order, dim3 grid, dim3 block, hipStream_t st, const Common& k, const float* states, int64_t Mn, float* out) {
    RSRL_VEV_CASE(0, 1) RSRL_VEV_CASE(0, 2) RSRL_VEV_CASE(0, 3) RSRL_VEV_CASE(0, 4) RSRL_VEV_CASE(0, 5) RSRL_VEV_CASE(1, 1) RSRL_VEV_CASE(2, 1)
    return false;
}
// TD / TDLambda / V-evaluate on the generic Fourier orders: states != nullptr -> evaluate, from != nullptr -> handle, else the driver loop
bool launch_td_model(const rsrl_hip_config& cfg, dim3 grid, dim3 block, hipStream_t st, const Common& k, const TdParams& tp, const BasisGeom& g, bool lambda,
                     uint64_t t, int chunk, DevStats* stats, const float* from, const float* rew, const float* to, const uint8_t* termf, int64_t Mn,
                     float* out, const float* states) {
#define RSRL_TDM_CASE(DM)                                                                                                                  \
    if (cfg.domain == DM) {                                                                                                                \
        using M = FourierGenericModel<DM>;                                                                                                 \
        if (states) hipLaunchKernelGGL((k_v_mem<M>), grid, block, 0, st, k, g, states, Mn, out);                                           \
        else hipLaunchKernelGGL((k_td_mem<M>), grid, block, 0, st, k, tp, g, lambda ? 1 : 0, t, chunk, stats, from, rew, to, termf, Mn, out); \
        return true;                                                                                                                       \
    }
    if (cfg.basis != RSRL_FOURIER || cfg.order < 1 || cfg.order > 7) return false;
    RSRL_TDM_CASE(0) RSRL_TDM_CASE(1) RSRL_TDM_CASE(2)
    return false;
}
bool launch_reset_td(int domain, dim3 grid, dim3 block, hipStream_t st, const Common& k, uint64_t t) {
    if (domain == 0) { hipLaunchKernelGGL((k_reset_td<0>), grid, block, 0, st, k, t); return true; }
    if (domain == 1) { hipLaunchKernelGGL((k_reset_td<1>), grid, block, 0, st, k, t); return true; }
    if (domain == 2) { hipLaunchKernelGGL((k_reset_td<2>), grid, block, 0, st, k, t); return true; }
    return false;
}
}  // namespace rsrl
