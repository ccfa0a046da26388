#!/usr/bin/env python3
"""Digest of what a set of configurations learn (weights, states, actions after a few hundred batch-steps), one line per configuration --
run under two builds of the library (RSRL_HIP_LIB) and diff the output: equal digests = the builds compute the same bits.
    python scripts/ab_bits.py > a.txt;  RSRL_HIP_LIB=$PWD/rsrl_amd/lib/<alt>.so python scripts/ab_bits.py > b.txt;  diff a.txt b.txt"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import rsrl_amd as ra  # noqa: E402

CASES = {
    "c2 qlearning fourier5": dict(n_envs=4096, policy=1, epsilon=0.1, gamma=0.9, lr=0.001, max_episode_steps=200),
    "sarsa fourier3": dict(n_envs=1024, order=3, algo=ra.SARSA, policy=1, epsilon=0.2, gamma=0.9, lr=0.002, max_episode_steps=100),
    "esarsa softmax cartpole": dict(n_envs=1024, domain=1, order=1, algo=ra.EXPECTED_SARSA, policy=ra.SOFTMAX, tau=0.5, gamma=0.95, lr=0.001, alpha=0.5,
                                    max_episode_steps=100),
    "sarsa-lambda fourier5": dict(n_envs=1024, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.1, gamma=0.99, alpha=0.001, lam=0.7, trace=1, max_episode_steps=100),
    "greedy-gq acrobot": dict(n_envs=512, domain=2, order=1, algo=ra.GREEDY_GQ, policy=1, epsilon=0.1, gamma=0.99, lr=0.01, lr_td=0.001, max_episode_steps=50),
    "c3 shared tiles": dict(n_envs=8192, domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA, policy=1, epsilon=0.1, gamma=0.99,
                            lr=0.0125 / 8192, weight_mode=ra.W_SHARED, max_episode_steps=100),
    "c4 shared fourier5": dict(n_envs=8192, policy=1, epsilon=0.1, gamma=0.9, lr=0.001 / 8192, weight_mode=ra.W_SHARED, max_episode_steps=100),
    "c5 wave bf16": dict(n_envs=256, domain=2, order=7, algo=ra.EXPECTED_SARSA, policy=ra.SOFTMAX, tau=1.0, gamma=0.99, lr=0.001, alpha=1.0,
                         weight_dtype=ra.W_BF16, max_episode_steps=50),
    "td-lambda tiles": dict(n_envs=64, domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.TD_LAMBDA, policy=ra.RANDOM, gamma=0.9, alpha=0.05,
                            lam=0.3, max_episode_steps=30),
}


def main():
    for name, kw in CASES.items():
        h = hashlib.sha256()
        with ra.Context(seed=5, **kw) as c:
            c.reset()
            for k in (97, 1, 158):
                c.train(k)
            h.update(np.ascontiguousarray(c.states).tobytes()); h.update(np.ascontiguousarray(c.actions).tobytes())
            shared = kw.get("weight_mode", 0) == ra.W_SHARED
            for i in ([0] if shared else range(0, c.N, max(1, c.N // 64))):
                h.update(np.ascontiguousarray(c.get_weights(i)).tobytes())
        print(f"{name:28s} {h.hexdigest()[:32]}")


if __name__ == "__main__":
    main()
