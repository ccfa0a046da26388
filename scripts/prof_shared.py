#!/usr/bin/env python3
"""One shared-W configuration run for a profiler: `fourier` (C4 share: 131 072 MountainCar envs) or `tile` (C3: 262 144 CartPole
envs), optionally with an exchange attached to the single rank (`rccl` / `peer`: a communicator of size 1 runs the multi-rank
sequence finalize -> exchange -> apply)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fourier"
exch = sys.argv[2] if len(sys.argv) > 2 else "none"
kw = dict(exchange=ra.EXCHANGE_PEER) if exch == "peer" else {}
if mode == "fourier":
    c = ra.Context(n_envs=131072, policy=1, epsilon=0.1, lr=0.001 / 131072, weight_mode=ra.W_SHARED, max_episode_steps=1000, **kw)
else:
    c = ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA, n_envs=262144, policy=1, epsilon=0.1,
                   gamma=0.99, lr=0.0125 / 262144, weight_mode=ra.W_SHARED, max_episode_steps=1000, **kw)
if exch == "rccl":
    c.comm_init(ra.Context.comm_unique_id(), 1, 0)
elif exch == "peer":
    c.peer_connect([c.peer_export(1)], 0)
c.reset(); c.train(64, want_stats=False); c.sync()
t0 = time.perf_counter(); c.train(320, want_stats=False); c.sync(); dt = time.perf_counter() - t0
print(mode, exch, "us/step %.2f" % (dt / 320 * 1e6), "env-steps/s %.3g" % (c.N * 320 / dt))
