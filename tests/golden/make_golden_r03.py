#!/usr/bin/env python3
"""Generates tests/golden/vectors_r03.json: golden vectors for what round 3 added -- the eligibility-trace and prediction agents
off the register family (tile coding, the order-7 wave family) and the shared-W dense step in the device's four-chain block order.
All of it is output of the CPU oracle in the device's arithmetic (f32d: restatement-derived; the reference is Rust and cannot run
here).  Bit patterns are stored as uint32, large matrices as position-weighted digests.
Run from the repo root:  python tests/golden/make_golden_r03.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).tolist()


def digest(a):
    b = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64).ravel()
    w = (2 * np.arange(b.size, dtype=np.uint64) + 1)
    return int((b * w).sum(dtype=np.uint64))


def record(run, st, n):
    return {"states": bits(run.state), "actions": run.action.tolist(), "episodes": st["episodes"], "episodes_truncated": st["episodes_truncated"],
            "w_digest": [digest(run.weights[i]) for i in range(n)], "z_digest": [digest(run.traces[i]) for i in range(n)]}


def main():
    out = {"oracle": {}}
    o = out["oracle"]
    # SARSA(lambda) on tile coding: CartPole, 4 tilings x 6^4, saturating traces, 6 learners x 60 batch-steps
    cfg = dict(domain=1, basis=orc.TILE, n_tilings=4, tiles_per_dim=6, algo=3, policy=1, epsilon=0.2, gamma=0.99, alpha=0.05, lam=0.8, trace=1,
               seed=7, max_episode_steps=19, env_offset=3)
    run = orc.Run(orc.make_agent(**cfg), 6, "f32d"); run.reset(); st = run.train(60)
    o["sarsa_lambda_tiles"] = {"config": cfg, "n_envs": 6, "steps": 60, **record(run, st, 6)}
    # TD(lambda) on tile coding: MountainCar, 8 tilings x 8^2, accumulating traces, random behaviour policy
    cfg = dict(domain=0, basis=orc.TILE, n_tilings=8, tiles_per_dim=8, algo=8, policy=3, gamma=0.9, alpha=0.05, lam=0.3, trace=0, seed=5,
               max_episode_steps=23)
    run = orc.Run(orc.make_agent(**cfg), 6, "f32d"); run.reset(); st = run.train(60)
    o["td_lambda_tiles"] = {"config": cfg, "n_envs": 6, "steps": 60, **record(run, st, 6)}
    # Q(lambda) on the wave family: Acrobot Fourier(7), Dutch traces, 3 learners x 24 batch-steps in the wave order
    cfg = dict(domain=2, order=7, algo=4, policy=1, epsilon=0.2, gamma=0.99, alpha=0.0005, lam=0.8, trace=2, seed=9, max_episode_steps=11)
    run = orc.Run(orc.make_agent(**cfg), 3, "f32d"); run.reset_wave(); st = run.train_wave(24)
    o["q_lambda_wave"] = {"config": cfg, "n_envs": 3, "steps": 24, **record(run, st, 3)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors_r03.json")
    json.dump(out, open(path, "w"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
