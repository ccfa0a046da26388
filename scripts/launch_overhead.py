#!/usr/bin/env python3
"""k_train_reg's cost per LAUNCH (prologue: 108 weight rows into registers; epilogue: back out) against its cost per batch-step:
HIP-event time of launches of 16 .. 4096 steps at BASELINE configs[1] (65 536 learners), least-squares line through them."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import rsrl_amd as ra  # noqa: E402


def main():
    pts = []
    for depth in (16, 64, 256, 1024, 4096):
        with ra.Context(n_envs=65536, policy=1, epsilon=0.1, gamma=0.9, lr=0.001, max_episode_steps=1000, steps_per_launch=depth) as c:
            c.reset()
            c.train(depth, want_stats=False); c.train(depth, want_stats=False)
            c.sync()
            c.timing_enable(True)
            calls = max(3, 8192 // depth)
            for _ in range(calls):
                c.train(depth, want_stats=False)
            c.sync()
            ms, n, kn = c.timing_read()
            pts.append((depth, ms * 1e3 / n, n, kn))
    d = np.array([p[0] for p in pts], float); t = np.array([p[1] for p in pts], float)
    slope, icpt = np.polyfit(d, t, 1)
    print(json.dumps({"kernel": pts[0][3], "us_per_launch_by_depth": {int(p[0]): p[1] for p in pts}, "us_per_batch_step": slope, "us_per_launch_overhead": icpt}))


if __name__ == "__main__":
    main()
