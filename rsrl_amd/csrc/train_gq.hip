// fused GreedyGQ driver loop + handle for the register family
#include "launch.hpp"
#include "kernels_gq.hpp"
#include "model_list.hpp"
namespace rsrl {

#define RSRL_GQ_CASE(DM, OR, PO)                                                                        \
    if (domain == DM && order == OR && policy == PO) {                                                  \
        hipLaunchKernelGGL((k_train_gq<DM, OR, PO>), grid, block, 0, st, k, gp, t, chunk, stats);       \
        return true;                                                                                    \
    }
#define RSRL_GQ_POLICIES(DM, OR) RSRL_GQ_CASE(DM, OR, 0) RSRL_GQ_CASE(DM, OR, 1) RSRL_GQ_CASE(DM, OR, 2) RSRL_GQ_CASE(DM, OR, 3)

bool launch_train_gq(int domain, int order, int policy, dim3 grid, dim3 block, hipStream_t st, const Common& k,
                     const GqParams& gp, uint64_t t, int chunk, DevStats* stats) {
    RSRL_GQ_POLICIES(0, 1) RSRL_GQ_POLICIES(0, 2) RSRL_GQ_POLICIES(0, 3) RSRL_GQ_POLICIES(0, 4) RSRL_GQ_POLICIES(0, 5)
    RSRL_GQ_POLICIES(1, 1) RSRL_GQ_POLICIES(2, 1)
    return false;
}
#define RSRL_HGQ_CASE(DM, OR)                                                                                       \
    if (domain == DM && order == OR) {                                                                              \
        hipLaunchKernelGGL((k_handle_gq<DM, OR>), grid, block, 0, st, k, gp, from, act, rew, to, termf, Mn, td_out); \
        return true;                                                                                                \
    }
bool launch_handle_gq(int domain, int order, dim3 grid, dim3 block, hipStream_t st, const Common& k, const GqParams& gp,
                      const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                      int64_t Mn, float* td_out) {
    RSRL_HGQ_CASE(0, 1) RSRL_HGQ_CASE(0, 2) RSRL_HGQ_CASE(0, 3) RSRL_HGQ_CASE(0, 4) RSRL_HGQ_CASE(0, 5) RSRL_HGQ_CASE(1, 1) RSRL_HGQ_CASE(2, 1)
    return false;
}
// GreedyGQ on the models without a register-family kernel (tile coding, generic Fourier orders): from == nullptr -> the driver loop
bool launch_gq_model(const rsrl_hip_config& cfg, dim3 grid, dim3 block, hipStream_t st, const Common& k, const GqParams& gp, const BasisGeom& g, uint64_t t,
                     int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                     int64_t Mn, float* td_out) {
#define X(TYPE, BS, DM, P)                                                                                                          \
    if (model_match(cfg, BS, DM, P)) {                                                                                               \
        using M = RSRL_UNPAREN TYPE;                                                                                                 \
        if (from) hipLaunchKernelGGL((k_handle_gq_mem<M>), grid, block, 0, st, k, gp, g, from, act, rew, to, termf, Mn, td_out);      \
        else hipLaunchKernelGGL((k_train_gq_mem<M>), grid, block, 0, st, k, gp, g, t, chunk, stats);                                 \
        return true;                                                                                                                 \
    }
    RSRL_MEM_MODELS(X)
#undef X
    return false;
}
}  // namespace rsrl
