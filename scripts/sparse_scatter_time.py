#!/usr/bin/env python3
"""k_sparse_trace_scatter timed bare at one size after a warm-up of W batch-steps (how full the lists are depends on it):
    [RSRL_SPARSE_CHUNK=512] python scripts/sparse_scatter_time.py <learners> [warm-up steps] [T]
one JSON line: us per batch-step of the scatter kernel (the ctx's timing hooks), mean live entries, GB/s on the live entries' bytes."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    kw = dict(domain=1, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=8, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.1, gamma=0.99, alpha=0.1 / T / n, lam=0.9,
              weight_mode=ra.W_SHARED, n_envs=n)
    with ra.Context(seed=1, max_episode_steps=200, **kw) as c:
        c.reset()
        c.train(warm, want_stats=False)
        c.sync()
        c.timing_enable(True)
        K = 32
        c.train(K, want_stats=False)
        c.sync()
        ms, _, kn = c.timing_read()
        us = ms * 1e3 / K
        live = float(np.mean([int((c.get_traces(i) != 0).sum()) for i in range(0, n, max(1, n // 64))]))
        real = 10 * live + 16 * 2 * 4 + 8 * (4 + 2 + 8)      # HBM bytes per learner-step (bench.py's accounting)
        print(json.dumps({"kernel": kn, "learners": n, "tilings": T, "warm": warm, "chunk": os.environ.get("RSRL_SPARSE_CHUNK", "auto"), "us_per_batch_step": round(us, 2),
                          "mean_live_entries": round(live, 1), "GBps_on_live_entries": round(real * n / us / 1e3, 1), "checksum": c.checksum()}), flush=True)


if __name__ == "__main__":
    main()
