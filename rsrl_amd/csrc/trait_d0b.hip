// trait-granular kernels (kernels_trait.hpp), MountainCar, Fourier order 5 (the bench's trait_loop leg)
#include "kernels_trait.hpp"
namespace rsrl {
bool launch_trait_lm_d0_high(int domain, int order, int algo, int policy, hipStream_t st, const Common& k, const TraitIo& io, uint64_t t) {
    RSRL_TRAIT_ALGOS(0, 5)
    return false;
}
}  // namespace rsrl
