"""rsrl_amd -- MI355X-native implementation of tspooner/rsrl's per-step TD-control hot path
(env.transition -> agent.handle -> policy.sample, rsrl/examples/q_learning.rs:34-55), vectorised over
N environments, behind the C ABI of include/rsrl_hip.h.  There is no CPU fallback: the HIP library
(rsrl_amd/lib/librsrl_hip.so, built by __graft_entry__.build()) must be present."""
from .context import (Q_SIGMA, GREEDY_GQ, PAL, TD, TD_LAMBDA, Q_LAMBDA, SARSA_LAMBDA, TRACE_ACCUMULATE, TRACE_DUTCH, TRACE_SATURATE, ACROBOT, CART_POLE, EPSILON_GREEDY, EXPECTED_SARSA, FOURIER, GREEDY, MOUNTAIN_CAR,  # noqa: F401
                      QLEARNING, RANDOM, SARSA, SOFTMAX, TILE_CODING, W_BF16, W_F32, W_PER_ENV, W_SHARED,
                      EXCHANGE_RCCL, EXCHANGE_PEER, EXCHANGE_AUTO, Context, device_count, device_identity, can_access_peer)
from ._abi import RsrlHipError  # noqa: F401

__all__ = ["Context", "RsrlHipError"]
