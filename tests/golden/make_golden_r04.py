#!/usr/bin/env python3
"""Generates tests/golden/vectors_r04.json: golden vectors for what round 4 added -- the per-learner epsilon schedule of the reference's
drivers (examples/sarsa_lambda.rs:48-75), Domain::rollout under a sampling policy (rsrl_domains/src/lib.rs:448-479), and GreedyGQ /
QSigma / TD off the register family (tile coding, generic Fourier orders).  All of it is output of the CPU oracle in the device's arithmetic
(f32d: restatement-derived; the reference is Rust and cannot run here).  Bit patterns are stored as uint32, matrices as position-weighted
digests.  Run from the repo root:  python tests/golden/make_golden_r04.py"""
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


def record(run, st, n, aux):
    r = {"states": bits(run.state), "actions": run.action.tolist(), "episodes": st["episodes"], "episodes_truncated": st["episodes_truncated"],
         "w_digest": [digest(run.weights[i]) for i in range(n)]}
    if aux:
        r["z_digest"] = [digest(run.traces[i]) for i in range(n)]
    return r


CASES = {
    # SARSA(lambda), the reference example's agent, epsilon *= 0.97 per episode of a learner with a floor
    "sarsa_lambda_eps_schedule": (dict(domain=0, order=5, algo=3, policy=1, trace=1, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.3, epsilon_decay=0.97,
                                       epsilon_min=0.05, seed=11, max_episode_steps=9), 8, 120, True),
    # CartPole episodes end at different steps for different learners: different epsilons
    "qlearning_eps_schedule": (dict(domain=1, order=1, algo=0, policy=1, gamma=0.9, lr=0.01, epsilon=0.5, epsilon_decay=0.9, seed=3, max_episode_steps=60), 8, 150, False),
    "greedy_gq_tiles": (dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=6, policy=1, gamma=0.99, lr=0.0125, lr_td=0.001, epsilon=0.1, seed=13,
                             max_episode_steps=15), 6, 80, True),
    "q_sigma_generic_fourier": (dict(domain=0, order=7, algo=9, policy=1, gamma=0.9, lr=0.01, alpha=0.5, sigma=0.5, n_steps=3, epsilon=0.2, seed=13,
                                     max_episode_steps=15), 6, 80, False),
    "td_generic_fourier": (dict(domain=1, order=2, algo=7, policy=3, gamma=0.9, lr=0.01, seed=13, max_episode_steps=15), 6, 80, False),
}


def main():
    out = {"oracle": {}}
    o = out["oracle"]
    for key, (cfg, n, steps, aux) in CASES.items():
        # the one-step agents on the register family run the device-order loop (carried phi / Q, rank-1 post-update Q: orc_run_train_dev);
        # every other kernel follows the reference's order of operations (orc_run_train)
        dev_loop = cfg["algo"] in (0, 1, 2, 5) and cfg.get("basis", 0) == 0
        run = orc.Run(orc.make_agent(**cfg), n, "f32d"); run.reset(); st = (run.train_dev if dev_loop else run.train)(steps)
        o[key] = {"config": cfg, "n_envs": n, "steps": steps, "aux": aux, "loop": "dev" if dev_loop else "hook", **record(run, st, n, aux)}
        if "epsilon_decay" in cfg:
            o[key]["eps_bits"] = bits(run.eps)
    # Domain::rollout under EpsilonGreedy(0.3) and Softmax(0.5) after 300 Q-learning steps: n_states, total reward, the actions taken
    cfg = dict(domain=0, order=3, algo=0, policy=1, gamma=0.9, lr=0.002, epsilon=0.2, seed=4, max_episode_steps=60)
    run = orc.Run(orc.make_agent(**cfg), 8, "f32d"); run.reset(); run.train_dev(300)
    roll = {"config": cfg, "n_envs": 8, "steps": 300, "limit": 40, "calls": []}
    for policy, kw in ((orc.EGREEDY, dict(epsilon=0.3)), (orc.SOFTMAX, dict(tau=0.5)), (orc.RANDOM, {})):
        n_states, total, acts = run.rollout_policy(policy, 40, call=len(roll["calls"]), **kw)
        roll["calls"].append({"policy": int(policy), "kw": kw, "n_states": n_states.tolist(), "total_reward_bits": bits(total), "actions": acts.tolist()})
    o["rollout_policy"] = roll
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors_r04.json")
    json.dump(out, open(path, "w"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
