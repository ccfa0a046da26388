#!/usr/bin/env python3
"""One configuration of bench.py run bare (no torch, no legs) for a profiler:  profile_leg.py <leg>

    fused      BASELINE configs[1]: 65 536 MountainCar, QLearning, Fourier(5), eps-greedy, per-env W; 256-step launches   k_train_reg
    stream     the same, one batch-step per launch, learner-major W (plain launches under RSRL_NO_GRAPH=1)                  k_step_reg_lm
    stream1m   the same at 1 048 576 learners: 453 MB of weights, beyond L2 (32 MB) and the Infinity Cache (256 MiB)         k_step_reg_q4
    persist    configs[3], one GPU's share: 131 072 MountainCar, ONE shared Fourier(5) approximator                         k_shared_persist
    perstep    the same, one launch per batch-step (RSRL_NO_PERSIST=1: the path of RCCL-attached ctxs)                      k_shared_step
    tile       configs[2]: 262 144 CartPole, SARSA, 8 x 8^4 tiles, one shared table                                         k_shared_ca, k_tile_scatter, k_apply_rep
    wave       configs[4], one GPU's share: 32 768 Acrobot, ExpectedSARSA, Fourier(7), Softmax, bf16 W                      k_train_wave_pk

Prints one JSON line: leg, kernel, learners, steps of the measured call, launches, HIP-event microseconds per batch-step."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402

# leg -> (config, batch-steps per train call, untimed calls, timed calls, batch-steps per DISPATCH of the leg's kernels)
LEGS = {
    "fused": (dict(n_envs=65536, policy=1, epsilon=0.1, gamma=0.9, lr=0.001, max_episode_steps=1000, steps_per_launch=256), 256, 2, 10, 256),
    "stream": (dict(n_envs=65536, policy=1, epsilon=0.1, gamma=0.9, lr=0.001, max_episode_steps=1000, steps_per_launch=1), 100, 1, 3, 1),
    "stream1m": (dict(n_envs=1048576, policy=1, epsilon=0.1, gamma=0.9, lr=0.001, max_episode_steps=1000, steps_per_launch=1), 20, 1, 3, 1),
    "persist": (dict(n_envs=131072, policy=1, epsilon=0.1, gamma=0.9, lr=0.001 / 131072, weight_mode=ra.W_SHARED, max_episode_steps=1000), 320, 2, 3, 320),
    # the same with RSRL_NO_PERSIST=1: one launch per batch-step (what RCCL-attached ctxs and shards beyond one block per CU run)
    "perstep": (dict(n_envs=131072, policy=1, epsilon=0.1, gamma=0.9, lr=0.001 / 131072, weight_mode=ra.W_SHARED, max_episode_steps=1000), 64, 1, 5, 1),
    "tile": (dict(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA, n_envs=262144, policy=1, epsilon=0.1, gamma=0.99,
                  lr=0.0125 / 262144, weight_mode=ra.W_SHARED, max_episode_steps=1000), 64, 1, 5, 1),
    "wave": (dict(domain=2, order=7, algo=ra.EXPECTED_SARSA, policy=ra.SOFTMAX, tau=1.0, gamma=0.99, lr=0.001, alpha=1.0, n_envs=32768,
                  weight_dtype=ra.W_BF16, max_episode_steps=1000), 64, 2, 4, 64),
}


def main():
    leg = sys.argv[1]
    kw, chunk, n_warm, n_calls, per_dispatch = LEGS[leg]
    if leg == "perstep":
        os.environ["RSRL_NO_PERSIST"] = "1"
    c = ra.Context(**kw)
    c.reset()
    for _ in range(n_warm):
        c.train(chunk, want_stats=False)
    c.sync()
    c.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(n_calls):
        c.train(chunk, want_stats=False)
    c.sync()
    dt = time.perf_counter() - t0
    ms, n, kn = c.timing_read()
    steps = chunk * n_calls
    print(json.dumps({"leg": leg, "kernel": kn, "learners": c.N, "steps_per_call": chunk, "warm_calls": n_warm, "calls": n_calls, "steps_per_dispatch": per_dispatch,
                      "launches": n, "wall_us_per_batch_step": dt / steps * 1e6, "event_us_per_batch_step": ms * 1e3 / steps,
                      "env_steps_per_s": c.N * steps / dt}), flush=True)
    c.close()


if __name__ == "__main__":
    main()
