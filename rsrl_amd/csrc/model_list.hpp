// model_list.hpp -- the (basis, domain, parameter) -> Model type table shared by the translation units that dispatch over it.
//   Fourier register family: F = (order+1)^D <= 36 features per learner held in VGPRs (FourierModel).
//   Tile coding: T tilings as a template parameter (indices in VGPRs), tiles_per_dim at run time (TileModel).
//   param -1: the generic-order Fourier model (any order 1..7 without a specialised kernel; listed last).
#pragma once

#include "../../include/rsrl_hip.h"
#include "models.hpp"

#define RSRL_MODELS(X)                                                                       \
    X((FourierModel<0, 1>), RSRL_FOURIER, 0, 1) X((FourierModel<0, 2>), RSRL_FOURIER, 0, 2)  \
    X((FourierModel<0, 3>), RSRL_FOURIER, 0, 3) X((FourierModel<0, 4>), RSRL_FOURIER, 0, 4)  \
    X((FourierModel<0, 5>), RSRL_FOURIER, 0, 5)                                              \
    X((FourierModel<1, 1>), RSRL_FOURIER, 1, 1) X((FourierModel<2, 1>), RSRL_FOURIER, 2, 1)  \
    X((TileModel<0, 4>), RSRL_TILE_CODING, 0, 4) X((TileModel<0, 8>), RSRL_TILE_CODING, 0, 8) X((TileModel<0, 16>), RSRL_TILE_CODING, 0, 16) \
    X((TileModel<1, 4>), RSRL_TILE_CODING, 1, 4) X((TileModel<1, 8>), RSRL_TILE_CODING, 1, 8) X((TileModel<1, 16>), RSRL_TILE_CODING, 1, 16) \
    X((TileModel<2, 4>), RSRL_TILE_CODING, 2, 4) X((TileModel<2, 8>), RSRL_TILE_CODING, 2, 8) X((TileModel<2, 16>), RSRL_TILE_CODING, 2, 16) \
    X((FourierGenericModel<0>), RSRL_FOURIER, 0, -1) X((FourierGenericModel<1>), RSRL_FOURIER, 1, -1) X((FourierGenericModel<2>), RSRL_FOURIER, 2, -1)
// the models WITHOUT a register-family kernel of their own (what the *_mem agent kernels are instantiated for)
#define RSRL_MEM_MODELS(X)                                                                   \
    X((TileModel<0, 4>), RSRL_TILE_CODING, 0, 4) X((TileModel<0, 8>), RSRL_TILE_CODING, 0, 8) X((TileModel<0, 16>), RSRL_TILE_CODING, 0, 16) \
    X((TileModel<1, 4>), RSRL_TILE_CODING, 1, 4) X((TileModel<1, 8>), RSRL_TILE_CODING, 1, 8) X((TileModel<1, 16>), RSRL_TILE_CODING, 1, 16) \
    X((TileModel<2, 4>), RSRL_TILE_CODING, 2, 4) X((TileModel<2, 8>), RSRL_TILE_CODING, 2, 8) X((TileModel<2, 16>), RSRL_TILE_CODING, 2, 16) \
    X((FourierGenericModel<0>), RSRL_FOURIER, 0, -1) X((FourierGenericModel<1>), RSRL_FOURIER, 1, -1) X((FourierGenericModel<2>), RSRL_FOURIER, 2, -1)

namespace rsrl {
template <class T> struct Tag { using type = T; };
#define RSRL_UNPAREN(...) __VA_ARGS__
static inline bool model_match(const rsrl_hip_config& cfg, int basis, int domain, int param) {
    if (cfg.basis != basis || cfg.domain != domain) return false;
    if (param == -1) return cfg.order >= 1 && cfg.order <= 7;
    return (basis == RSRL_FOURIER ? cfg.order : cfg.n_tilings) == param;
}
}  // namespace rsrl
