#!/usr/bin/env python3
"""Worst-case device-vs-f64-oracle errors of the single-step quantities of SURVEY 8(d) (phi, Q, delta, W after one update,
the three transitions), on many random in-range states.  Prints one JSON object; tests/test_gpu_parity_mc.py asserts ~2x these
figures and bench.py reports its own (smaller) live sample as `parity`."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (test infrastructure: the checker)


def rand_states(domain, M, seed):
    lo, hi = orc.domain_bounds(domain)
    rng = np.random.default_rng(seed)
    return (lo[:, None] + (hi - lo)[:, None] * rng.random((len(lo), M))).astype(np.float32)


def measure(M=4096, seed=0):
    out = {}
    s = rand_states(0, M, seed)
    with ra.Context(n_envs=M, algo=ra.QLEARNING, policy=ra.GREEDY, gamma=0.95, lr=0.05) as c:
        phi = c.project(s)
        out["phi_max_abs"] = float(max(np.abs(phi[:, m] - orc.fourier_project(0, 5, s[:, m], "f64")).max() for m in range(M)))
        rng = np.random.default_rng(seed + 1)
        Ws = (rng.normal(size=(M, 36, 3)) * 0.5).astype(np.float32)
        for i in range(M):
            c.set_weights(Ws[i], i)
        ag = orc.make_agent(gamma=0.95, lr=0.05)
        q = c.q_evaluate(s)
        qe = 0.0
        for i in range(M):
            q64 = orc.q_evaluate(ag, Ws[i].astype(np.float64), s[:, i], "f64")
            qe = max(qe, float(np.abs(q[:, i] - q64).max() / (1 + np.abs(q64).max())))
        out["q_max_rel"] = qe
        a = rng.integers(0, 3, M).astype(np.int32)
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        td = c.handle(frm, a, rew, nxt, term)
        de = we = 0.0
        for i in range(M):
            W = Ws[i].astype(np.float64).copy()
            d = orc.handle(ag, W, frm[:, i], a[i], rew[i], nxt[:, i], term[i], orc.draw(0, i, 0, orc.BLK_INNER), "f64")
            de = max(de, abs(td[i] - d) / (1 + abs(d)))
            we = max(we, float(np.max(np.abs(c.get_weights(i) - W))) / (1 + abs(d)))
        out["delta_max_rel"] = float(de)
        out["w_update_max_abs_rel"] = float(we)
    for domain, name in ((0, "mountain_car"), (1, "cart_pole"), (2, "acrobot")):
        s = rand_states(domain, M, seed + 10 + domain)
        if domain == 1:
            s *= 0.5
        rng = np.random.default_rng(seed + 20 + domain)
        with ra.Context(domain=domain, order=5 if domain == 0 else 1, n_envs=M) as c:
            c.states = s
            a = rng.integers(0, c.A, M).astype(np.int32)
            _, nxt, _, _ = c.domain_step(a)
            e = 0.0
            for i in range(M):
                es, _, _ = orc.domain_step(domain, s[:, i], a[i], "f64")
                e = max(e, float(np.max(np.abs(nxt[:, i] - es) / (1 + np.abs(es)))))
            out[f"step_{name}_max_rel"] = e
    return out


if __name__ == "__main__":
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)))
