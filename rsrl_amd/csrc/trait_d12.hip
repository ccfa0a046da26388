// trait-granular kernels (kernels_trait.hpp), CartPole / Acrobot, Fourier order 1; the dispatcher; policy.sample
#include "kernels_trait.hpp"
namespace rsrl {
bool launch_trait_lm_d0_low(int domain, int order, int algo, int policy, hipStream_t st, const Common& k, const TraitIo& io, uint64_t t);
bool launch_trait_lm_d0_high(int domain, int order, int algo, int policy, hipStream_t st, const Common& k, const TraitIo& io, uint64_t t);
static bool launch_trait_lm_d12(int domain, int order, int algo, int policy, hipStream_t st, const Common& k, const TraitIo& io, uint64_t t) {
    RSRL_TRAIT_ALGOS(1, 1) RSRL_TRAIT_ALGOS(2, 1)
    return false;
}
bool trait_lm_available(int domain, int order, int algo) {
    const bool basis = (domain == 0 && (order == 1 || order == 3 || order == 5)) || ((domain == 1 || domain == 2) && order == 1);
    return basis && (algo == 0 || algo == 1 || algo == 2 || algo == 5);
}
bool launch_trait_lm(int domain, int order, int algo, int policy, hipStream_t st, const Common& k, const TraitIo& io, uint64_t t) {
    if (!trait_lm_available(domain, order, algo)) return false;
    if (domain == 0) return order == 5 ? launch_trait_lm_d0_high(domain, order, algo, policy, st, k, io, t) : launch_trait_lm_d0_low(domain, order, algo, policy, st, k, io, t);
    return launch_trait_lm_d12(domain, order, algo, policy, st, k, io, t);
}
bool launch_trait_sample(int domain, int order, hipStream_t st, const Common& k, const float* states, int64_t Mn, uint64_t t, uint32_t blk, float* qkey,
                         int32_t* actions_out) {
    RSRL_TRAIT_SAMPLE_CASE(0, 1) RSRL_TRAIT_SAMPLE_CASE(0, 3) RSRL_TRAIT_SAMPLE_CASE(0, 5) RSRL_TRAIT_SAMPLE_CASE(1, 1) RSRL_TRAIT_SAMPLE_CASE(2, 1)
    return false;
}
}  // namespace rsrl
