#!/usr/bin/env python3
"""The kernels round 5 added inside SURVEY 8's rows, timed bare (HIP events of the ctx's timing hooks) at sizes that fill the device:
    python scripts/profile_round5_kernels.py            -> one JSON line per kernel: learners, us per batch-step, env-steps/s, algorithmic GB/s
Algorithmic bytes per learner-step (what the formulation must move; 4-byte values):
    wave-family agents (F = 4096, one wavefront per learner, W streamed):  GreedyGQ  read W (3F) + V (3F), write 2 columns of W + 1 of V;
        TD read w (F) write w (F); TDLambda read w + z, write w + z; QSigma read W (3F) + write 1 column + the ring
    sparse-trace lambda agents over a shared table: a learner's list read + written (<= 512 x 8 B each way) + 2 x T gathers + <= 512 atomics
    lambda agents on a generic Fourier order: W and Z read and written (2 x 2 x F x A)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402

F7 = 4096
CASES = [
    ("k_wave_aux GreedyGQ", dict(domain=2, order=7, algo=ra.GREEDY_GQ, policy=1, epsilon=0.1, gamma=0.99, lr=1e-4, lr_td=1e-4, n_envs=8192), (6 * F7 + 3 * F7) * 4),
    ("k_wave_aux TD", dict(domain=2, order=7, algo=ra.TD, policy=ra.RANDOM, gamma=0.99, lr=1e-4, n_envs=8192), 2 * F7 * 4),
    ("k_wave_aux TDLambda", dict(domain=1, order=7, algo=ra.TD_LAMBDA, policy=ra.RANDOM, gamma=0.9, lam=0.5, n_envs=8192), 4 * F7 * 4),
    ("k_wave_qsigma", dict(domain=2, order=7, algo=ra.Q_SIGMA, policy=1, epsilon=0.1, gamma=0.99, lr=1e-4, alpha=0.5, sigma=0.5, n_steps=4, n_envs=8192), (3 * F7 + F7) * 4),
    ("k_wave_lambda SARSALambda", dict(domain=2, order=7, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.1, gamma=0.99, alpha=1e-4, lam=0.8, n_envs=8192), 4 * 3 * F7 * 4),
    ("k_sparse_trace_scatter", dict(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.1, gamma=0.99,
                                  alpha=0.1 / 8 / 16384, lam=0.9, weight_mode=ra.W_SHARED, n_envs=16384), 2 * 512 * 8 + 16 * 2 * 4 + 512 * 8),
    ("k_train_lambda_mem", dict(domain=1, order=3, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.1, gamma=0.99, alpha=1e-3, lam=0.8, n_envs=16384), 4 * 256 * 2 * 4),
    # (16 384 learners x 4 threads are ONE wave per SIMD; the BASELINE configurations' 65 536 learners fill the device)
    ("k_train_lambda_mem 65536", dict(domain=1, order=3, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.1, gamma=0.99, alpha=1e-3, lam=0.8, n_envs=65536), 4 * 256 * 2 * 4),
    ("k_sparse_trace_scatter 65536", dict(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.1, gamma=0.99,
                                          alpha=0.1 / 8 / 65536, lam=0.9, weight_mode=ra.W_SHARED, n_envs=65536), 2 * 512 * 8 + 16 * 2 * 4 + 512 * 8),
    ("k_lambda_tile", dict(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.1, gamma=0.99, alpha=0.01, lam=0.8,
                           n_envs=1024), 4 * 32768 * 2 * 4),
]


def main():
    for name, kw, bytes_per in CASES:
        with ra.Context(seed=1, max_episode_steps=200, **kw) as c:
            c.reset()
            c.train(8, want_stats=False)
            c.sync()
            c.timing_enable(True)
            K = 32
            c.train(K, want_stats=False)
            c.sync()
            ms, n, kn = c.timing_read()
            us = ms * 1e3 / K
            n_envs = kw["n_envs"]
            extra = {}
            if kn == "k_sparse_trace_scatter":       # what the lists really hold: HBM bytes on the LIVE entries (16-bit key + value read, value written: 10 B; bench.py's accounting)
                import numpy as np
                live = float(np.mean([int((c.get_traces(i) != 0).sum()) for i in range(0, n_envs, max(1, n_envs // 64))]))
                bytes_per = round(10 * live + 16 * 2 * 4 + 8 * (4 + 2 + 8))
                extra = {"mean_live_entries": round(live, 1), "full_list_formula_bytes": 12416}
            print(json.dumps({"what": name, **extra, "kernel": kn, "learners": n_envs, "us_per_batch_step": round(us, 3), "env_steps_per_s": round(n_envs / us * 1e6, 1),
                              "algorithmic_bytes_per_learner_step": bytes_per, "algorithmic_GBps": round(bytes_per * n_envs / us / 1e3, 1),
                              "frac_of_8TBps": round(bytes_per * n_envs / us / 1e3 / 8000, 3)}), flush=True)


if __name__ == "__main__":
    main()
