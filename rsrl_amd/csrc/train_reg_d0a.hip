// fused train kernels, MountainCar, Fourier orders 1-4
#include "launch.hpp"
namespace rsrl {
bool launch_train_reg_d0_low(int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                             const Common& k, uint64_t t, int chunk, DevStats* stats, const uint64_t* t_dev) {
    RSRL_TRAIN_ALGOS(0, 1) RSRL_TRAIN_ALGOS(0, 2) RSRL_TRAIN_ALGOS(0, 3) RSRL_TRAIN_ALGOS(0, 4)
    return false;
}
}  // namespace rsrl
