#!/usr/bin/env python3
"""The trait-granular loop with HOST arrays (numpy through the ctypes binding): every call stages its arrays over PCIe and synchronises -- the PCIe-inclusive
rate DESIGN.md §5 quotes beside the device-pointer loop's (scripts/trait_loop.py).  Never part of bench.py's `value`.
    python scripts/trait_loop_host.py [n_envs=65536] [steps=200]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    with ra.Context(n_envs=n, policy=ra.EPSILON_GREEDY, epsilon=0.1, gamma=0.9, lr=0.001, max_episode_steps=0, steps_per_launch=1) as c:
        c.reset()
        act = c.actions

        def step(a):
            frm, to, rew, term = c.domain_step(a)
            c.handle(frm, a, rew, to, term)
            c.domain_reset(term)
            return c.policy_sample(c.states)

        for _ in range(20):
            act = step(act)
        t0 = time.perf_counter()
        for _ in range(steps):
            act = step(act)
        dt = time.perf_counter() - t0
        bytes_per_step = n * (4 + 2 * 8 + 4 + 1) * 2 + n * (1 + 8 + 8 + 4)      # domain_step out + handle in; reset mask, states out + in, actions out
        print(json.dumps({"what": "trait loop, HOST arrays (PCIe + one synchronisation per call)", "learners": n, "steps": steps, "us_per_batch_step": dt / steps * 1e6,
                          "env_steps_per_s": n * steps / dt, "pcie_bytes_per_batch_step": bytes_per_step, "pcie_GBps": bytes_per_step * steps / dt / 1e9}))


if __name__ == "__main__":
    main()
