// fused train kernels, CartPole, Fourier order 1
#include "launch.hpp"
namespace rsrl {
bool launch_train_reg_d1(int order, int algo, int policy, dim3 grid, dim3 block, hipStream_t st,
                         const Common& k, uint64_t t, int chunk, DevStats* stats, const uint64_t* t_dev) {
    RSRL_TRAIN_ALGOS(1, 1)
    return false;
}
}  // namespace rsrl
