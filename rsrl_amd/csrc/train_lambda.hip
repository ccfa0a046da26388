// fused eligibility-trace driver loops (SARSA(lambda), Q(lambda)) for the register family
#include "launch.hpp"
#include "kernels_lambda.hpp"
#include "kernels_lambda_mem.hpp"
#include <cstdlib>
namespace rsrl {

#define RSRL_LAMBDA_CASE(DM, OR, AL, PO)                                                                      \
    if (domain == DM && order == OR && algo == AL && policy == PO) {                                          \
        hipLaunchKernelGGL((k_train_lambda<DM, OR, AL, PO>), grid, block, 0, st, k, lp, t, chunk, stats);     \
        return true;                                                                                          \
    }
#define RSRL_LAMBDA_POLICIES(DM, OR, AL) \
    RSRL_LAMBDA_CASE(DM, OR, AL, 0) RSRL_LAMBDA_CASE(DM, OR, AL, 1) RSRL_LAMBDA_CASE(DM, OR, AL, 2) RSRL_LAMBDA_CASE(DM, OR, AL, 3)
#define RSRL_LAMBDA_ALGOS(DM, OR) RSRL_LAMBDA_POLICIES(DM, OR, 3) RSRL_LAMBDA_POLICIES(DM, OR, 4)

bool launch_train_lambda(int domain, int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                         const Common& k, const LambdaParams& lp, uint64_t t, int chunk, DevStats* stats) {
    RSRL_LAMBDA_ALGOS(0, 1) RSRL_LAMBDA_ALGOS(0, 2) RSRL_LAMBDA_ALGOS(0, 3) RSRL_LAMBDA_ALGOS(0, 4) RSRL_LAMBDA_ALGOS(0, 5)
    RSRL_LAMBDA_ALGOS(1, 1) RSRL_LAMBDA_ALGOS(2, 1)
    return false;
}
#define RSRL_HL_CASE(DM, OR)                                                                                              \
    if (domain == DM && order == OR) {                                                                                    \
        hipLaunchKernelGGL((k_handle_lambda<DM, OR>), grid, block, 0, st, k, lp, from, act, rew, to, termf, Mn, t, td_out); \
        return true;                                                                                                      \
    }
bool launch_handle_lambda(int domain, int order, dim3 grid, dim3 block, hipStream_t st, const Common& k, const LambdaParams& lp,
                          const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                          int64_t Mn, uint64_t t, float* td_out) {
    RSRL_HL_CASE(0, 1) RSRL_HL_CASE(0, 2) RSRL_HL_CASE(0, 3) RSRL_HL_CASE(0, 4) RSRL_HL_CASE(0, 5) RSRL_HL_CASE(1, 1) RSRL_HL_CASE(2, 1)
    return false;
}
// SARSALambda / QLambda on the generic Fourier orders (kernels_lambda_mem.hpp): from != nullptr -> handle, else the driver loop
bool launch_lambda_model(const rsrl_hip_config& cfg, dim3 grid, dim3 block, hipStream_t st, const Common& k, const LambdaParams& lp, const BasisGeom& g,
                         uint64_t t, int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to,
                         const uint8_t* termf, int64_t Mn, float* td_out) {
#define RSRL_LM_CASE(DM)                                                                                                             \
    if (cfg.domain == DM) {                                                                                                          \
        using M = FourierGenericModel<DM>;                                                                                           \
        if (from) hipLaunchKernelGGL((k_handle_lambda_mem<M>), grid, block, 0, st, k, lp, g, from, act, rew, to, termf, Mn, t, td_out); \
        else if (!getenv("RSRL_LAMBDA_MEM1"))        /* four threads per learner: 64 learners per block (RSRL_LAMBDA_MEM1=1: the one-thread form, A/B) */ \
            hipLaunchKernelGGL((k_train_lambda_mem4<M>), dim3((unsigned)((k.n_envs + 63) / 64)), dim3(256), 0, st, k, lp, g, t, chunk, stats);                \
        else hipLaunchKernelGGL((k_train_lambda_mem<M>), grid, block, 0, st, k, lp, g, t, chunk, stats);                             \
        return true;                                                                                                                 \
    }
    if (cfg.basis != RSRL_FOURIER || cfg.order < 1 || cfg.order > 7) return false;
    RSRL_LM_CASE(0) RSRL_LM_CASE(1) RSRL_LM_CASE(2)
    return false;
}
}  // namespace rsrl
