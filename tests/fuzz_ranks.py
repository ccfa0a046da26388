#!/usr/bin/env python3
"""Multi-rank campaign on ONE device: G in-process ranks (one ctx and one host thread each, peer-write exchange through same-process pointers) over ONE
shared approximator, random agent / basis / learner count (ragged shards included) / step split -- every replica of W must be identical, and the group
must reproduce the unsharded run: bit for bit when every shard is whole 512-learner blocks of the dense basis or the basis is tile coding with an exact
fixed-point sum per rank (then only the G-term float sum regroups: <= 2e-9 absolute at |W| ~ 1e-3), to the rounding of regrouped block sums otherwise.

    GPU_MAX_HW_QUEUES=32 python tests/fuzz_ranks.py [n_cases=40] [seed=0]      (the HIP runtime reads the variable at start-up: more queues than ranks,
                                                                                 so that a waiting kernel never sits in front of a peer's)"""
import json
import os
import sys
import threading

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402
from rsrl_amd.distributed import shard_range  # noqa: E402


def run_all(ctxs, calls):
    out, errs = [None] * len(ctxs), []

    def work(r):
        try:
            c = ctxs[r]
            c.reset()
            for k in calls:
                c.train(k, want_stats=False)
            c.sync()
            out[r] = (c.get_weights().copy(), c.states.copy(), c.actions.copy())
        except Exception as e:      # noqa: BLE001
            errs.append(f"rank {r}: {e!r}"[:300])
    th = [threading.Thread(target=work, args=(r,)) for r in range(len(ctxs))]
    [t.start() for t in th]
    [t.join(180) for t in th]
    if any(t.is_alive() for t in th):
        errs.append("a rank did not return within 180 s")
    return out, errs


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    failures, counts = [], {}
    for idx in range(n_cases):
        fam = str(rng.choice(["dense", "dense", "tile", "sparse_lambda"]))
        G = int(rng.choice([2, 3, 4, 8]))
        domain = int(rng.integers(0, 3))
        if fam == "dense":
            N = int(rng.choice([G * 512, 2 * G * 512, 1000, 4173, 700])) if G <= 4 else int(rng.choice([G * 512, 4000, 4173]))
            kw = dict(domain=domain, order=int(rng.integers(1, 6)) if domain == 0 else 1, algo=int(rng.integers(0, 3)), lr=float(rng.choice([0.01, 0.1])) / N)
        else:
            N = int(rng.choice([64 * G, 1000, 2048, 777]))
            kw = dict(domain=domain, basis=ra.TILE_CODING, n_tilings=int(rng.choice([4, 8, 16])), tiles_per_dim=int(rng.choice([4, 8])))
            if fam == "tile":
                kw.update(algo=int(rng.integers(0, 3)), lr=0.1 / kw["n_tilings"] / N)
            else:
                kw.update(algo=int(rng.choice([ra.SARSA_LAMBDA, ra.Q_LAMBDA])), alpha=0.1 / kw["n_tilings"] / N, lam=float(rng.choice([0.5, 0.9])), trace=int(rng.integers(0, 3)))
        kw.update(policy=int(rng.choice([0, 1, 1, 2])), epsilon=0.2, tau=1.0, gamma=float(rng.choice([0.9, 0.99])), weight_mode=ra.W_SHARED,
                  seed=int(rng.integers(0, 1 << 20)), max_episode_steps=int(rng.choice([0, 30, 200])), exchange=ra.EXCHANGE_PEER)
        calls = [int(rng.integers(3, 30))] + ([int(rng.integers(1, 40))] if rng.random() < 0.6 else [])
        tag = f"{idx:3d} {fam:13s} G {G} N {N:5d} dom {domain} algo {kw['algo']} calls {calls}"
        try:
            ctxs = [ra.Context(n_envs=cnt, env_offset=off, **kw) for off, cnt in (shard_range(N, G, r) for r in range(G))]
            handles = [c.peer_export(G) for c in ctxs]
            for r, c in enumerate(ctxs):
                c.peer_connect(handles, r)
        except ra.RsrlHipError as e:
            counts["refused"] = counts.get("refused", 0) + 1
            print(tag + f"  REFUSED {str(e)[:100]}", flush=True)
            continue
        out, errs = run_all(ctxs, calls)
        bad = list(errs)
        if not errs:
            for r in range(1, G):
                if not np.array_equal(out[0][0], out[r][0], equal_nan=True):
                    bad.append(f"replica {r} of W differs from replica 0")
            with ra.Context(n_envs=N, **dict(kw, exchange=ra.EXCHANGE_AUTO)) as full:
                full.reset()
                for k in calls:
                    full.train(k, want_stats=False)
                ref = (full.get_weights().copy(), full.states.copy(), full.actions.copy())
            absw = float(np.abs(ref[0]).max())
            err_w = float(np.max(np.abs(ref[0] - out[0][0])))
            states = np.concatenate([o[1] for o in out], axis=1)
            same = float(np.all(states == ref[1], axis=0).mean())
            whole = fam == "dense" and N % (G * 512) == 0
            if whole and (err_w != 0.0 or same != 1.0):
                bad.append(f"whole-block shards: err_w {err_w:.2e}, same {same:.3f}")
            # (regrouped block sums differ in their rounding; once that tips a learner's argmax its trajectory -- and from there the shared W -- parts
            #  by whole updates: seen once in 570 cases, 0.3 % of 4 173 learners after 54 steps, W off by 5e-5 at |W| = 1)
            if not whole and (same < 0.97 or err_w > (2e-6 if same == 1.0 else 1e-3) * max(1.0, absw)):
                bad.append(f"err_w {err_w:.2e} at |W| {absw:.2e}, same {same:.3f}")
            tag += f"  err_w {err_w:.1e} |W| {absw:.1e} same {same:.3f}"
        for c in ctxs:
            c.close()
        status = "MISMATCH" if bad else "ok"
        counts[status] = counts.get(status, 0) + 1
        print(tag + ("  " + status + (" " + str(bad) if bad else "")), flush=True)
        if bad:
            failures.append({"case": idx, "line": tag, "bad": bad, "config": {k: (v if not isinstance(v, np.generic) else v.item()) for k, v in kw.items()}, "G": G, "N": N})
    print("SUMMARY " + json.dumps({"cases": n_cases, "seed": seed, "counts": counts, "failures": failures}, default=str), flush=True)
    os._exit(1 if failures else 0)


if __name__ == "__main__":
    main()
