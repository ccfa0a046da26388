#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration that fits one GPU (parity-test configs, not bench lines):
prints one JSON object per config with env-steps/s and the dominant kernel's average launch time."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402

CONFIGS = {
    "C2 65536 MountainCar QL Fourier(5) eps-greedy per-env W (fused)": (dict(n_envs=65536, policy=1, epsilon=0.1, max_episode_steps=1000), 5120, 512, 608),
    "C2 same, 1 step per launch": (dict(n_envs=65536, policy=1, epsilon=0.1, max_episode_steps=1000, steps_per_launch=1), 3000, 300, 608),
    "L1 65536 MountainCar SARSA(lambda) Fourier(5) replacing traces (examples/sarsa_lambda.rs)": (dict(n_envs=65536, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.2, gamma=0.99, alpha=0.01,
                                                                                                     lam=0.7, trace=ra.TRACE_SATURATE, max_episode_steps=1000), 2560, 256, 1040),
    "L2 2048 CartPole SARSA(lambda) tiles 8x8^4, per-learner dense trace tables (1 MiB of Z + W swept per learner-step)": (dict(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA_LAMBDA,
                                                                 n_envs=2048, policy=1, epsilon=0.2, gamma=0.99, alpha=0.0125, lam=0.7, trace=ra.TRACE_SATURATE, max_episode_steps=1000), 128, 16, 1048576 + 208),
    "L3 2048 Acrobot SARSA(lambda) Fourier(7) wave family, f32 W + Z streamed (240 KiB per learner-step)": (dict(domain=2, order=7, algo=ra.SARSA_LAMBDA, n_envs=2048, policy=1, epsilon=0.2, gamma=0.99,
                                                                 alpha=0.0005, lam=0.8, trace=ra.TRACE_SATURATE, max_episode_steps=1000), 64, 8, 5 * 49152 + 64),
    "L3b 16384 Acrobot SARSA(lambda) Fourier(7) wave family (1.6 GB of W + Z: HBM-resident)": (dict(domain=2, order=7, algo=ra.SARSA_LAMBDA, n_envs=16384, policy=1, epsilon=0.2, gamma=0.99,
                                                                 alpha=0.0005, lam=0.8, trace=ra.TRACE_SATURATE, max_episode_steps=1000), 32, 4, 5 * 49152 + 64),
    "P1 2048 CartPole TD(lambda) tiles 8x8^4, per-learner dense trace (256 KiB of Z + W swept per learner-step)": (dict(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.TD_LAMBDA,
                                                                 n_envs=2048, policy=ra.RANDOM, gamma=0.9, alpha=0.05, lam=0.3, max_episode_steps=1000), 128, 16, 4 * 131072 + 64),
    "K1 262144 MountainCar QL Fourier(5), 1 step per launch (four lanes per learner)": (dict(n_envs=262144, policy=1, epsilon=0.1, max_episode_steps=1000, steps_per_launch=1), 2000, 300, 608),
    "C3 262144 CartPole SARSA tiles 8x8^4 shared W": (dict(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA, n_envs=262144, policy=1,
                                                          epsilon=0.1, gamma=0.99, lr=0.0125 / 262144, weight_mode=ra.W_SHARED, max_episode_steps=1000), 256, 64, 208),
    "C3' 16384 CartPole SARSA tiles 8x8^4 per-env W (4 GiB of tables)": (dict(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA, n_envs=16384, policy=1,
                                                                              epsilon=0.1, gamma=0.99, lr=0.0125, max_episode_steps=1000, steps_per_launch=64), 512, 64, 208),
    "C4/8 131072 MountainCar shared-W QL Fourier(5) (one GPU's share)": (dict(n_envs=131072, policy=1, epsilon=0.1, lr=0.001 / 131072, weight_mode=ra.W_SHARED, max_episode_steps=1000), 320, 64, 32),
    "C5/2 32768 Acrobot ExpectedSARSA Fourier(7) Softmax bf16 W (one GPU's share)": (dict(domain=2, order=7, algo=ra.EXPECTED_SARSA, policy=ra.SOFTMAX, tau=1.0, gamma=0.99, lr=0.001, alpha=1.0,
                                                                                          n_envs=32768, weight_dtype=ra.W_BF16, max_episode_steps=1000, steps_per_launch=64), 256, 64, 32816),
    "C5' 32768 Acrobot ExpectedSARSA Fourier(7) Softmax f32 W": (dict(domain=2, order=7, algo=ra.EXPECTED_SARSA, policy=ra.SOFTMAX, tau=1.0, gamma=0.99, lr=0.001, alpha=1.0,
                                                                      n_envs=32768, weight_dtype=ra.W_F32, max_episode_steps=1000, steps_per_launch=64), 256, 64, 65584),
}

only = sys.argv[1:] or None
for name, (kw, steps, warm, bytes_per_step) in CONFIGS.items():
    if only and not any(o in name for o in only):
        continue
    try:
        c = ra.Context(**kw)
        c.reset()
        c.train(warm, want_stats=False)
        c.sync()
        c.timing_enable(True)
        t0 = time.perf_counter()
        c.train(steps, want_stats=False)
        c.sync()
        dt = time.perf_counter() - t0
        ms, n, kn = c.timing_read()
        st = c.train(max(1, steps // 8))
        rec = {"config": name, "env_steps_per_s": c.N * steps / dt, "us_per_batch_step": dt / steps * 1e6, "kernel": kn,
               "avg_launch_us": ms * 1e3 / max(1, n), "launches": n,
               "algorithmic_GBps": bytes_per_step * c.N * steps / (ms * 1e-3) / 1e9 if ms > 0 else None,
               "frac_of_8TBps": bytes_per_step * c.N * steps / (ms * 1e-3) / 8e12 if ms > 0 else None,
               "episodes_per_1k_env_steps": 1000.0 * st["episodes"] / st["env_steps"]}
        c.close()
    except Exception as e:
        rec = {"config": name, "error": repr(e)}
    print(json.dumps(rec), flush=True)
