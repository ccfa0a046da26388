// trait-granular kernels (kernels_trait.hpp), MountainCar, Fourier orders 1 and 3
#include "kernels_trait.hpp"
namespace rsrl {
bool launch_trait_lm_d0_low(int domain, int order, int algo, int policy, hipStream_t st, const Common& k, const TraitIo& io, uint64_t t) {
    RSRL_TRAIT_ALGOS(0, 1) RSRL_TRAIT_ALGOS(0, 3)
    return false;
}
}  // namespace rsrl
