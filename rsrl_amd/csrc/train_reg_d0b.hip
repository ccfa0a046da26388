// fused train kernels, MountainCar, Fourier order 5 (the headline configuration)
#include "launch.hpp"
namespace rsrl {
bool launch_train_reg_d0_low(int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                             const Common& k, uint64_t t, int chunk, DevStats* stats, const uint64_t* t_dev);
bool launch_train_reg_d0(int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                         const Common& k, uint64_t t, int chunk, DevStats* stats, const uint64_t* t_dev) {
    RSRL_TRAIN_ALGOS(0, 5)
    return launch_train_reg_d0_low(order, algo, policy, grid, block, st, k, t, chunk, stats, t_dev);
}
}  // namespace rsrl
