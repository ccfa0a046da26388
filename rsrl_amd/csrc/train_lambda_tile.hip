// eligibility-trace control on tile coding, per-learner tables: one block per learner (kernels_lambda_tile.hpp)
#include "launch.hpp"
#include "kernels_lambda_tile.hpp"
#include "kernels_td_tile.hpp"
namespace rsrl {

#define RSRL_TT_CASE(DM, TT)                                                                                                              \
    if (domain == DM && n_tilings == TT) {                                                                                                \
        if (eval_states) hipLaunchKernelGGL((k_v_tile<DM, TT>), dim3((unsigned)((Mn + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, k, g, eval_states, Mn, td_out); \
        else hipLaunchKernelGGL((k_td_tile<DM, TT, 256>), dim3((unsigned)n_blocks), dim3(256), 0, st, k, g, tp, lambda ? 1 : 0, t, chunk, stats, from, rew, \
                                to, termf, Mn, td_out);                                                                                   \
        return true;                                                                                                                      \
    }
// TD / TDLambda on tile coding.  eval_states != nullptr: V(s) of Mn states into td_out; from != nullptr: handle; else the driver loop
bool launch_td_tile(int domain, int n_tilings, bool lambda, int64_t n_blocks, hipStream_t st, const Common& k, const BasisGeom& g, const TdParams& tp,
                    uint64_t t, int chunk, DevStats* stats, const float* from, const float* rew, const float* to, const uint8_t* termf, int64_t Mn,
                    float* td_out, const float* eval_states) {
    RSRL_TT_CASE(0, 4) RSRL_TT_CASE(0, 8) RSRL_TT_CASE(0, 16)
    RSRL_TT_CASE(1, 4) RSRL_TT_CASE(1, 8) RSRL_TT_CASE(1, 16)
    RSRL_TT_CASE(2, 4) RSRL_TT_CASE(2, 8) RSRL_TT_CASE(2, 16)
    return false;
}

#define RSRL_LT_CASE(DM, TT)                                                                                                          \
    if (domain == DM && n_tilings == TT) {                                                                                            \
        hipLaunchKernelGGL((k_lambda_tile<DM, TT, 256>), dim3((unsigned)n_blocks), dim3(256), 0, st, k, g, lp, t, chunk, stats, from, act, rew, \
                           to, termf, Mn, td_out);                                                                                    \
        return true;                                                                                                                  \
    }
bool launch_lambda_tile(int domain, int n_tilings, int64_t n_blocks, hipStream_t st, const Common& k, const BasisGeom& g, const LambdaParams& lp,
                        uint64_t t, int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to,
                        const uint8_t* termf, int64_t Mn, float* td_out) {
    RSRL_LT_CASE(0, 4) RSRL_LT_CASE(0, 8) RSRL_LT_CASE(0, 16)
    RSRL_LT_CASE(1, 4) RSRL_LT_CASE(1, 8) RSRL_LT_CASE(1, 16)
    RSRL_LT_CASE(2, 4) RSRL_LT_CASE(2, 8) RSRL_LT_CASE(2, 16)
    return false;
}
}  // namespace rsrl
