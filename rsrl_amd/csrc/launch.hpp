// launch.hpp -- launchers of the fused train kernels, one translation unit per domain so the
// (order x algo x policy) instantiations compile in parallel.
#pragma once
#include "kernels_reg.hpp"
#include "kernels_reg_q4.hpp"

namespace rsrl {

// returns false when no instantiation exists for (order, algo, policy)
bool launch_train_reg_d0(int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                         const Common& k, uint64_t t, int chunk, DevStats* stats, const uint64_t* t_dev = nullptr);
bool launch_train_reg_d1(int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                         const Common& k, uint64_t t, int chunk, DevStats* stats, const uint64_t* t_dev = nullptr);
bool launch_train_reg_d2(int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                         const Common& k, uint64_t t, int chunk, DevStats* stats, const uint64_t* t_dev = nullptr);

// chunk == -1 selects the single-step streaming kernel (k_step_reg), -2 its learner-major form (k_step_reg_lm), -3 the learner-major
// form with four lanes per learner (k_step_reg_q4: grid = learners / 64)
struct LambdaParams;
struct BasisGeom;
bool launch_train_lambda(int domain, int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                         const Common& k, const LambdaParams& lp, uint64_t t, int chunk, DevStats* stats);
bool launch_handle_lambda(int domain, int order, dim3 grid, dim3 block, hipStream_t st, const Common& k, const LambdaParams& lp,
                          const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                          int64_t Mn, uint64_t t, float* td_out);

// lambda agents on tile coding, per-learner tables: n_blocks = learners (driver loop) or Mn (handle: from != nullptr)
bool launch_lambda_tile(int domain, int n_tilings, int64_t n_blocks, hipStream_t st, const Common& k, const BasisGeom& g, const LambdaParams& lp,
                        uint64_t t, int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to,
                        const uint8_t* termf, int64_t Mn, float* td_out);

struct TdParams;
bool launch_td_tile(int domain, int n_tilings, bool lambda, int64_t n_blocks, hipStream_t st, const Common& k, const BasisGeom& g, const TdParams& tp,
                    uint64_t t, int chunk, DevStats* stats, const float* from, const float* rew, const float* to, const uint8_t* termf, int64_t Mn,
                    float* td_out, const float* eval_states);

struct GqParams;
bool launch_train_gq(int domain, int order, int policy, dim3 grid, dim3 block, hipStream_t st, const Common& k,
                     const GqParams& gp, uint64_t t, int chunk, DevStats* stats);
bool launch_handle_gq(int domain, int order, dim3 grid, dim3 block, hipStream_t st, const Common& k, const GqParams& gp,
                      const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                      int64_t Mn, float* td_out);

struct TdParams;
bool launch_train_td(int domain, int order, bool lambda, dim3 grid, dim3 block, hipStream_t st, const Common& k, const TdParams& tp,
                     uint64_t t, int chunk, DevStats* stats);
bool launch_handle_td(int domain, int order, bool lambda, dim3 grid, dim3 block, hipStream_t st, const Common& k, const TdParams& tp,
                      const float* from, const float* rew, const float* to, const uint8_t* termf, int64_t Mn, float* td_out);
bool launch_v_evaluate(int domain, int order, dim3 grid, dim3 block, hipStream_t st, const Common& k, const float* states, int64_t Mn, float* out);
bool launch_reset_td(int domain, dim3 grid, dim3 block, hipStream_t st, const Common& k, uint64_t t);

struct QsParams;
struct BasisGeom;
// from == nullptr: the QSigma driver loop (chunk batch-steps); otherwise Handler::handle on Mn caller-supplied transitions
bool launch_qsigma(int domain, int order, dim3 grid, dim3 block, hipStream_t st, const Common& k, const QsParams& qp, const BasisGeom& g, uint64_t t,
                   int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                   int64_t Mn, float* td_out);

// the same two agents on the models without a register-family kernel (tile coding, generic Fourier orders); false if the configuration
// has a register-family kernel (use the launchers above) or none at all
}  // namespace rsrl
#include "../../include/rsrl_hip.h"
namespace rsrl {
bool launch_gq_model(const rsrl_hip_config& cfg, dim3 grid, dim3 block, hipStream_t st, const Common& k, const GqParams& gp, const BasisGeom& g, uint64_t t,
                     int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                     int64_t Mn, float* td_out);
bool launch_lambda_model(const rsrl_hip_config& cfg, dim3 grid, dim3 block, hipStream_t st, const Common& k, const LambdaParams& lp, const BasisGeom& g,
                         uint64_t t, int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to,
                         const uint8_t* termf, int64_t Mn, float* td_out);
bool launch_td_model(const rsrl_hip_config& cfg, dim3 grid, dim3 block, hipStream_t st, const Common& k, const TdParams& tp, const BasisGeom& g, bool lambda,
                     uint64_t t, int chunk, DevStats* stats, const float* from, const float* rew, const float* to, const uint8_t* termf, int64_t Mn,
                     float* out, const float* states);
bool launch_qsigma_model(const rsrl_hip_config& cfg, dim3 grid, dim3 block, hipStream_t st, const Common& k, const QsParams& qp, const BasisGeom& g, uint64_t t,
                         int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                         int64_t Mn, float* td_out);

#define RSRL_TRAIN_CASE(DM, OR, AL, PO)                                                                     \
    if (order == OR && algo == AL && policy == PO) {                                                        \
        if (chunk == -3) {                                                                                  \
            if constexpr (FourierReg<DM, OR>::F % 4 == 0 && Domain<DM>::A <= 3)                               \
                hipLaunchKernelGGL((k_step_reg_q4<DM, OR, AL, PO>), dim3((unsigned)((k.n_envs + 63) / 64)), block, 0, st, k, t, stats, t_dev); \
            else return false;                                                                              \
        } else if (chunk == -2) {                                                                           \
            if constexpr ((Domain<DM>::A * FourierReg<DM, OR>::F) % 4 == 0 && FourierReg<DM, OR>::F % 4 == 0)     \
                hipLaunchKernelGGL((k_step_reg_lm<DM, OR, AL, PO>), grid, block, 0, st, k, t, stats, t_dev);       \
            else return false;                                                                              \
        } else if (chunk == -1)                                                                             \
            hipLaunchKernelGGL((k_step_reg<DM, OR, AL, PO>), grid, block, 0, st, k, t, stats, t_dev);              \
        else if (PO == POL_EGREEDY && k.eps) {      /* per-learner epsilon schedule: an instantiation of its own (EpsilonGreedy only) */ \
            if constexpr (PO == POL_EGREEDY) hipLaunchKernelGGL((k_train_reg<DM, OR, AL, PO, true>), grid, block, 0, st, k, t, chunk, stats); \
        } else                                                                                              \
            hipLaunchKernelGGL((k_train_reg<DM, OR, AL, PO>), grid, block, 0, st, k, t, chunk, stats);            \
        return true;                                                                                        \
    }
#define RSRL_TRAIN_POLICIES(DM, OR, AL) \
    RSRL_TRAIN_CASE(DM, OR, AL, 0) RSRL_TRAIN_CASE(DM, OR, AL, 1) RSRL_TRAIN_CASE(DM, OR, AL, 2) RSRL_TRAIN_CASE(DM, OR, AL, 3)
#define RSRL_TRAIN_ALGOS(DM, OR) RSRL_TRAIN_POLICIES(DM, OR, 0) RSRL_TRAIN_POLICIES(DM, OR, 1) RSRL_TRAIN_POLICIES(DM, OR, 2) RSRL_TRAIN_POLICIES(DM, OR, 5)

}  // namespace rsrl
