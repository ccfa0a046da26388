// QSigma kernels (register-family Fourier bases, weights in memory)
#include "launch.hpp"
#include "kernels_qsigma.hpp"
#include "model_list.hpp"
namespace rsrl {
#define RSRL_QS_CASE(DM, OR)                                                                                                       \
    if (domain == DM && order == OR) {                                                                                             \
        using M = FourierModel<DM, OR>;                                                                                            \
        if (from) hipLaunchKernelGGL((k_handle_qsigma<M>), grid, block, 0, st, k, qp, g, from, act, rew, to, termf, Mn, t, td_out);  \
        else hipLaunchKernelGGL((k_train_qsigma<M>), grid, block, 0, st, k, qp, g, t, chunk, stats);                               \
        return true;                                                                                                               \
    }
// from == nullptr: the driver loop (chunk batch-steps); otherwise Handler::handle on Mn caller-supplied transitions
bool launch_qsigma(int domain, int order, dim3 grid, dim3 block, hipStream_t st, const Common& k, const QsParams& qp, const BasisGeom& g, uint64_t t,
                   int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                   int64_t Mn, float* td_out) {
    RSRL_QS_CASE(0, 1) RSRL_QS_CASE(0, 2) RSRL_QS_CASE(0, 3) RSRL_QS_CASE(0, 4) RSRL_QS_CASE(0, 5) RSRL_QS_CASE(1, 1) RSRL_QS_CASE(2, 1)
    return false;
}
// ... and on every other model (tile coding with per-learner tables, the generic Fourier orders): the agent is generic over the
// approximator (q_sigma.rs:80-105), and so are k_train_qsigma / k_handle_qsigma
bool launch_qsigma_model(const rsrl_hip_config& cfg, dim3 grid, dim3 block, hipStream_t st, const Common& k, const QsParams& qp, const BasisGeom& g, uint64_t t,
                         int chunk, DevStats* stats, const float* from, const int32_t* act, const float* rew, const float* to, const uint8_t* termf,
                         int64_t Mn, float* td_out) {
#define X(TYPE, BS, DM, P)                                                                                                          \
    if (model_match(cfg, BS, DM, P)) {                                                                                               \
        using M = RSRL_UNPAREN TYPE;                                                                                                 \
        if (from) hipLaunchKernelGGL((k_handle_qsigma<M>), grid, block, 0, st, k, qp, g, from, act, rew, to, termf, Mn, t, td_out);   \
        else hipLaunchKernelGGL((k_train_qsigma<M>), grid, block, 0, st, k, qp, g, t, chunk, stats);                                 \
        return true;                                                                                                                 \
    }
    RSRL_MEM_MODELS(X)
#undef X
    return false;
}
}  // namespace rsrl
