#!/usr/bin/env python3
"""Generates tests/golden/vectors.json: small input/output vectors for the hot path.

Two kinds (labelled in the file):
  "reference": constants transcribed from the reference's own unit tests (rsrl_domains/src/cart_pole.rs:143-183,
               policies/greedy.rs:96-168, epsilon_greedy.rs:115-145, softmax.rs:273-291 ...): they do not depend on
               this script at all and are what pins the oracle.
  "oracle":    outputs of the CPU oracle (f64 instantiation) for inputs chosen here -- the reference is Rust and
               cannot be built or run in this image (no cargo/rustc; crate lfa absent), so these are
               restatement-derived and labelled as such.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def main():
    rng = np.random.default_rng(20260928)
    out = {"reference": {}, "oracle": {}}
    out["reference"]["cartpole_steps_action0"] = [
        [-0.0032931628891235, -0.3293940797883472, 0.0029499634056967, 0.2951522145037250],
        [-0.0131819582085161, -0.6597158115002169, 0.0118185373734479, 0.5921703414056713]]
    out["reference"]["greedy_argmax"] = {"q": [-123.1, 123.1, 250.5, -1240.0, -4500.0, 10000.0, 20.1], "a": 5}
    out["reference"]["egreedy_probs"] = {"eps": 0.5, "cases": [
        {"q": [1.0, 0.0, 0.0, 0.0, 0.0], "p": [0.6, 0.1, 0.1, 0.1, 0.1]},
        {"q": [1.0, 0.0, 0.0, 0.0, 1.0], "p": [0.35, 0.1, 0.1, 0.1, 0.35]}]}
    o = out["oracle"]
    # C.3: Fourier-5 phi(-0.5, 0)
    o["fourier5_phi_default_state"] = orc.fourier_project(0, 5, [-0.5, 0.0]).tolist()
    # domain steps from random fp32-representable states
    o["domain_steps"] = []
    for dom in (0, 1, 2):
        lo, hi = orc.domain_bounds(dom)
        for _ in range(12):
            s = (lo + (hi - lo) * (0.25 + 0.5 * rng.random(len(lo)))).astype(np.float32)
            a = int(rng.integers(0, orc.lib().orc_domain_actions(dom)))
            ns, r, term = orc.domain_step(dom, s.astype(np.float64), a)
            o["domain_steps"].append({"domain": dom, "s": s.tolist(), "a": a, "ns": ns.tolist(), "r": r, "term": term})
    # tile indices for a 16-state CartPole sample (integer: bit-exact)
    ag = orc.make_agent(domain=1, basis=orc.TILE, n_tilings=8, tiles_per_dim=8)
    lo, hi = orc.domain_bounds(1)
    o["cartpole_tile_indices"] = []
    for _ in range(16):
        s = (lo + (hi - lo) * rng.random(4)).astype(np.float32)
        o["cartpole_tile_indices"].append({"s": s.tolist(), "idx": orc.tile_indices(ag, s).tolist()})
    # one update of each agent from a fixed non-zero W
    W0 = (rng.normal(size=(36, 3)) * 0.2).astype(np.float32)
    o["W0"] = W0.tolist()
    o["updates"] = []
    for algo, policy in ((0, 0), (1, 1), (2, 1), (2, 2)):
        kw = dict(gamma=0.95, lr=0.05, alpha=0.5, epsilon=0.2, tau=0.8)
        agx = orc.make_agent(algo=algo, policy=policy, seed=5, **kw)
        s = np.array([-0.6, 0.01], dtype=np.float32)
        ns, r, term = orc.domain_step(0, s.astype(np.float64), 2)
        ns = ns.astype(np.float32)
        x_in = orc.draw(5, 0, 0, orc.BLK_INNER)
        W = W0.astype(np.float64).copy()
        d = orc.handle(agx, W, s, 2, r, ns, term, x_in)
        o["updates"].append({"algo": algo, "policy": policy, **kw, "s": s.tolist(), "a": 2, "r": r, "ns": ns.tolist(),
                             "term": term, "delta": d, "W_col2_after": W[:, 2].tolist()})
    # 1000-step teacher-forced W and greedy-rollout n_states for fixed W
    agt = orc.make_agent(policy=orc.EGREEDY, epsilon=0.1, seed=8, gamma=0.9, lr=0.01, max_episode_steps=200)
    run = orc.Run(agt, 4, "f64")
    run.reset()
    trans = []
    for k in range(1000):
        s, a = run.state.copy(), run.action.copy()
        step = []
        for i in range(4):
            ns, r, t = orc.domain_step(0, s[i], a[i])
            step.append({"s": s[i].astype(np.float32).tolist(), "a": int(a[i]), "r": r, "ns": ns.astype(np.float32).tolist(), "term": t})
        trans.append(step)
        run.train(1)
    o["teacher_forced"] = {"config": dict(policy=1, epsilon=0.1, seed=8, gamma=0.9, lr=0.01), "transitions": trans,
                           "W_after": run.weights.tolist()}
    agr = orc.make_agent(policy=orc.EGREEDY, epsilon=0.1, seed=2, gamma=0.99, lr=0.005, max_episode_steps=400)
    run = orc.Run(agr, 16, "f64")
    run.reset()
    run.train(3000)
    W32 = run.weights.astype(np.float32)
    run.weights[:] = W32
    n, tot = run.rollout_greedy(500)
    o["greedy_rollout"] = {"W": W32.tolist(), "limit": 500, "n_states": n.tolist(), "total_reward": tot.tolist()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json")
    json.dump(out, open(path, "w"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
