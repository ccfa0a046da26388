#!/usr/bin/env python3
"""Generates tests/golden/vectors_r02.json: golden vectors for what round 2 added -- the device-order (f32d) runs the HIP path
must reproduce BIT FOR BIT, and QSigma.  All of it is output of the CPU oracle ("oracle": restatement-derived, the reference is
Rust and cannot run here; QSigma additionally carries the documented one-line repair).  Bit patterns are stored as uint32.
Run from the repo root:  python tests/golden/make_golden_r02.py"""
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


def main():
    out = {"oracle": {}}
    o = out["oracle"]
    # C2, device order: MountainCar QLearning Fourier(5) eps-greedy, 32 learners x 500 batch-steps
    cfg = dict(policy=1, epsilon=0.1, gamma=0.9, lr=0.001, seed=3, max_episode_steps=120)
    run = orc.Run(orc.make_agent(**cfg), 32, "f32d")
    run.reset()
    st = run.train_dev(500)
    o["c2_device_order"] = {"config": cfg, "n_envs": 32, "steps": 500, "states": bits(run.state), "actions": run.action.tolist(),
                            "w_learner0": bits(run.weights[0]), "w_digest_all": digest(run.weights), "episodes": st["episodes"]}
    # C5, wave order: Acrobot ExpectedSARSA Fourier(7) Softmax, bf16 weights, 4 learners x 40 batch-steps
    cfg5 = dict(domain=2, order=7, algo=2, policy=2, gamma=0.99, lr=0.001, alpha=1.0, tau=1.0, seed=23, max_episode_steps=30)
    run = orc.Run(orc.make_agent(**cfg5), 4, "f32d")
    run.reset_wave()
    st = run.train_wave(40, bf16=True)
    o["c5_wave_order_bf16"] = {"config": cfg5, "n_envs": 4, "steps": 40, "states": bits(run.state), "actions": run.action.tolist(),
                               "w_digest": [digest(run.weights[i]) for i in range(4)], "w0_first_rows": bits(run.weights[0][:4]),
                               "episodes": st["episodes"]}
    # QSigma: 2 learners, 10 teacher-forced transitions each, n_steps = 3, sigma = 0.5
    rng = np.random.default_rng(20260929)
    lo, hi = orc.domain_bounds(0)
    cfgq = dict(algo=9, policy=1, sigma=0.5, n_steps=3, gamma=0.9, lr=0.05, alpha=0.6, epsilon=0.2, seed=4)
    ag = orc.make_agent(**cfgq)
    W0 = (rng.normal(size=(2, 36, 3)) * 0.2).astype(np.float32)
    Wd, W6 = W0.copy(), W0.astype(np.float64)
    bd = [orc.QSigmaBackup(3, "f32d") for _ in range(2)]
    b6 = [orc.QSigmaBackup(3, "f64") for _ in range(2)]
    steps = []
    s = (lo[:, None] + (hi - lo)[:, None] * rng.random((2, 2))).astype(np.float32)
    for k in range(10):
        a = rng.integers(0, 3, 2).astype(np.int32)
        ns = (lo[:, None] + (hi - lo)[:, None] * rng.random((2, 2))).astype(np.float32)
        term = np.array([k == 6, k == 8], dtype=np.uint8)
        res_d, res_6 = [], []
        for i in range(2):
            x = orc.draw(4, i, k, orc.BLK_INNER)
            res_d.append(bd[i].handle(ag, Wd[i], s[:, i], a[i], -1.0, ns[:, i], term[i], x))
            res_6.append(b6[i].handle(ag, W6[i], s[:, i], a[i], -1.0, ns[:, i], term[i], x))
        steps.append({"s": s.tolist(), "a": a.tolist(), "ns": ns.tolist(), "term": term.tolist(), "residual_f32d_bits": bits(res_d), "residual_f64": res_6})
        s = ns
    o["qsigma"] = {"config": cfgq, "W0": W0.tolist(), "steps": steps, "W_after_f32d_bits": bits(Wd), "W_after_f64": W6.tolist()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors_r02.json")
    json.dump(out, open(path, "w"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
