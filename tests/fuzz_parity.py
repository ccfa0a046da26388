#!/usr/bin/env python3
"""Differential campaign: RANDOM configurations of the whole supported space, the HIP path (through the C ABI) against the oracle's
device-order instantiation ("f32d"), bit for bit -- states, actions, every learner's weights and auxiliary matrix -- over a short run cut
into random train() calls.  The fixed test-suite pins chosen configurations; this samples the combinations nobody chose
(domain x basis x order x agent x policy x weight mode x dtype x fuse depth x episode cap x learner count x env offset).

    python tests/fuzz_parity.py [n_cases=200] [seed=0]          (GPU box; test infrastructure: imports oracle/)

Prints one line per case and a JSON summary; exit code 1 if any accepted configuration differs from the oracle."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ONE_STEP = (ra.QLEARNING, ra.SARSA, ra.EXPECTED_SARSA, ra.PAL)
LAMBDA = (ra.SARSA_LAMBDA, ra.Q_LAMBDA)
PRED = (ra.TD, ra.TD_LAMBDA)
NAMES = {0: "QLearning", 1: "SARSA", 2: "ExpectedSARSA", 3: "SARSALambda", 4: "QLambda", 5: "PAL", 6: "GreedyGQ", 7: "TD", 8: "TDLambda", 9: "QSigma"}
N_DIM = {0: 2, 1: 4, 2: 4}


def sample(rng):
    """-> (family, device kwargs, oracle kwargs, oracle method name, method kwargs)"""
    domain = int(rng.integers(0, 3))
    family = rng.choice(["reg", "reg", "generic", "tile", "tile", "wave", "shared_dense", "shared_tile", "sparse_lambda"])
    kw = dict(domain=domain, seed=int(rng.integers(0, 1 << 20)), gamma=float(rng.choice([0.0, 0.9, 0.9, 0.99, 1.0])),
              max_episode_steps=int(rng.choice([0, 1, 25, 25, 200])), env_offset=int(rng.choice([0, 0, 64, 1000003])))
    n = int(rng.choice([1, 3, 63, 64, 65, 130, 257, 257, 1000, 2049]))
    method, mkw = "train", {}
    if family in ("reg", "shared_dense"):
        kw.update(basis=ra.FOURIER, order=int(rng.integers(1, 6)) if domain == 0 else 1)
    elif family == "generic":
        kw.update(basis=ra.FOURIER, order=int(rng.choice([6, 7])) if domain == 0 else int(rng.choice([2, 3, 4])))
    elif family == "wave":
        if domain == 0:
            domain = kw["domain"] = int(rng.integers(1, 3))
        kw.update(basis=ra.FOURIER, order=7)
        n = int(rng.choice([1, 3, 4, 5, 9, 16]))
    else:
        kw.update(basis=ra.TILE_CODING, n_tilings=int(rng.choice([4, 8, 16])), tiles_per_dim=int(rng.choice([4, 6, 8])))
    feats = (kw.get("order", 0) + 1) ** N_DIM[domain] if kw["basis"] == ra.FOURIER else kw["n_tilings"]
    step = float(rng.choice([0.02, 0.2])) / feats                 # lr |phi|^2 well below 1
    # ---- the agent
    if family == "shared_dense" or family == "shared_tile":
        algo = int(rng.choice(ONE_STEP))
    elif family == "sparse_lambda":
        algo = int(rng.choice(LAMBDA))
    else:
        algo = int(rng.integers(0, 10))
    kw["algo"] = algo
    kw["policy"] = ra.RANDOM if algo in PRED else int(rng.choice([ra.GREEDY, ra.EPSILON_GREEDY, ra.EPSILON_GREEDY, ra.SOFTMAX]))
    kw["epsilon"] = float(rng.choice([0.0, 0.05, 0.3, 0.3, 1.0]))
    kw["tau"] = float(rng.choice([0.05, 0.5, 1.0, 5.0]))
    kw["lr"] = step
    kw["alpha"] = float(rng.choice([0.5, 1.0])) if algo in (ra.EXPECTED_SARSA, ra.PAL) else step
    if algo in LAMBDA + (ra.TD_LAMBDA,):
        kw.update(lam=float(rng.choice([0.0, 0.5, 0.9, 0.9, 1.0])), trace=int(rng.integers(0, 3)))
    if algo == ra.GREEDY_GQ:
        kw["lr_td"] = step * float(rng.choice([0.1, 1.0]))
    if algo == ra.Q_SIGMA:
        kw.update(sigma=float(rng.choice([0.0, 0.5, 1.0])), n_steps=int(rng.choice([1, 2, 5])), alpha=float(rng.choice([0.3, 1.0])) * step * feats)
    if algo in (ra.SARSA, ra.EXPECTED_SARSA, ra.SARSA_LAMBDA) and rng.random() < 0.3:     # the agent's own policy object
        kw.update(agent_policy=int(rng.choice([ra.GREEDY, ra.EPSILON_GREEDY, ra.SOFTMAX])), agent_epsilon=0.2, agent_tau=2.0)
    dev = dict(kw, n_envs=n)
    if family in ("tile", "generic", "wave") or (family == "reg" and algo not in ONE_STEP):
        dev["steps_per_launch"] = int(rng.choice([0, 0, 1, 5]))
    # ---- which loop of the oracle restates the kernel family's evaluation order
    if family == "reg" and algo in ONE_STEP:
        method = "train_dev"
        dev["steps_per_launch"] = int(rng.choice([0, 0, 1, 7]))
        if algo != ra.PAL and kw["policy"] == ra.EPSILON_GREEDY and dev["steps_per_launch"] != 1 and rng.random() < 0.25:
            kw.update(epsilon_decay=0.97, epsilon_min=0.01)
            dev.update(epsilon_decay=0.97, epsilon_min=0.01)
    elif family in ("reg", "tile", "generic") and kw["policy"] == ra.EPSILON_GREEDY and dev.get("steps_per_launch", 0) != 1 and rng.random() < 0.25 and \
            ((family == "reg" and algo in LAMBDA) or (family != "reg" and algo in ONE_STEP)):
        kw.update(epsilon_decay=0.97, epsilon_min=0.01)           # the drivers' per-episode schedule (examples/sarsa_lambda.rs:68)
        dev.update(epsilon_decay=0.97, epsilon_min=0.01)
    elif family == "wave":
        method = "train_wave"
        if (algo in ONE_STEP[:3] or algo in LAMBDA or algo in (ra.GREEDY_GQ, ra.TD, ra.TD_LAMBDA, ra.Q_SIGMA)) and rng.random() < 0.5:      # bf16 + stochastic rounding (the trace / aux agents: round 6)
            dev["weight_dtype"] = ra.W_BF16
            mkw["bf16"] = True
    elif family == "shared_dense":
        method = "train_shared_dev"
        dev["weight_mode"] = ra.W_SHARED
        n = dev["n_envs"] = int(rng.choice([64, 512, 600, 1024, 1500]))
        dev["lr"] = kw["lr"] = step / n
    elif family == "shared_tile":
        dev["weight_mode"] = ra.W_SHARED
        n = dev["n_envs"] = int(rng.choice([64, 300, 1024, 2000]))
        dev["lr"] = kw["lr"] = step / n
    elif family == "sparse_lambda":
        method = "train_sparse_lambda"
        dev["weight_mode"] = ra.W_SHARED
        n = dev["n_envs"] = int(rng.choice([5, 64, 300]))
        dev["alpha"] = kw["alpha"] = step / n
    okw = {k: v for k, v in kw.items() if k not in ("basis",)}
    okw["basis"] = orc.TILE if kw["basis"] == ra.TILE_CODING else orc.FOURIER
    okw["shared_w"] = dev.get("weight_mode", ra.W_PER_ENV) == ra.W_SHARED
    if okw["max_episode_steps"] == 0:
        okw["max_episode_steps"] = 0
    return family, dev, okw, method, mkw


def run_case(rng, idx):
    family, dev, okw, method, mkw = sample(rng)
    while os.environ.get("FUZZ_FAMILY") and family != os.environ["FUZZ_FAMILY"]:      # one family only (e.g. after a change to its kernels)
        family, dev, okw, method, mkw = sample(rng)
    n = dev["n_envs"]
    total = int(rng.choice([40, 90, 150])) if family != "wave" else int(rng.choice([8, 20, 60]))
    if os.environ.get("FUZZ_LONG"):      # long horizons: across the default fuse depth (4 096), many graph replays, thousands of episodes
        total *= 10 if family == "wave" else 60
        if n > 130 and family not in ("shared_dense", "shared_tile"):
            n = dev["n_envs"] = 130
        elif n > 600:
            n = dev["n_envs"] = 600
    cuts = sorted(set(int(x) for x in rng.integers(1, total, size=int(rng.integers(0, 3)))))
    calls = [b - a for a, b in zip([0] + cuts, cuts + [total])]
    if rng.random() < 0.1:
        calls.insert(int(rng.integers(0, len(calls) + 1)), 0)     # a call of zero batch-steps changes nothing
    tag = f"{idx:4d} {family:13s} {NAMES[dev['algo']]:13s} dom {dev['domain']} N {n:5d} K {total:3d} calls {calls}"
    try:
        ctx = ra.Context(**dev)
    except ra.RsrlHipError as e:
        return "refused", tag + f"  REFUSED: {str(e)[:90]}", dev
    with ctx as c:
        ag = orc.make_agent(**okw)
        run = orc.Run(ag, n, "f32d")
        try:
            (run.reset_wave if method == "train_wave" else run.reset)()
            c.reset()
            stat_bad = None
            for k in calls:
                ost = getattr(run, method)(k, **mkw) if k else None
                dst = c.train(k, want_stats=bool(rng.integers(0, 2)))
                # the call's statistics: counters exact, the f64 sums of fp32 terms to their summation order
                if ost and dst and stat_bad is None:
                    for key in ("env_steps", "episodes", "episodes_truncated", "sum_episode_steps"):
                        if int(ost[key]) != int(dst[key]):
                            stat_bad = f"stats.{key}: device {dst[key]} oracle {ost[key]}"
                    for key in ("sum_abs_td_error", "sum_reward"):
                        a_, b_ = float(dst[key]), float(ost[key])
                        if np.isfinite(a_) and np.isfinite(b_) and abs(a_ - b_) > 2e-4 * (1 + abs(b_)):
                            stat_bad = stat_bad or f"stats.{key}: device {a_} oracle {b_}"
        except ValueError as e:                                   # the oracle has no loop for it
            return "no_oracle", tag + f"  NO ORACLE LOOP: {str(e)[:80]}", dev
        bad = [stat_bad] if stat_bad else []
        if not np.array_equal(c.states.T, run.state, equal_nan=True):
            bad.append("states")
        if not np.array_equal(c.actions, run.action):
            bad.append("actions")
        shared = dev.get("weight_mode", ra.W_PER_ENV) == ra.W_SHARED
        ow = run.weights
        if shared:
            if not np.array_equal(c.get_weights(), ow.reshape(c.get_weights().shape), equal_nan=True):
                bad.append("weights")
        else:
            for i in sorted(set([0, n // 2, n - 1])):
                if not np.array_equal(c.get_weights(i), ow[i].reshape(c.get_weights(i).shape), equal_nan=True):
                    bad.append(f"weights[{i}]")
        if family == "sparse_lambda":
            for i in sorted(set([0, n - 1])):
                if not np.array_equal(c.get_traces(i), run.sparse_trace(i)):
                    bad.append(f"sparse trace[{i}]")
        elif dev["algo"] in LAMBDA + (ra.TD_LAMBDA, ra.GREEDY_GQ):
            get = c.get_td_weights if dev["algo"] == ra.GREEDY_GQ else c.get_traces
            for i in sorted(set([0, n - 1])):
                if not np.array_equal(get(i), run.traces[i].reshape(get(i).shape), equal_nan=True):
                    bad.append(f"aux[{i}]")
        finite = bool(np.all(np.isfinite(ow)))
        moved = bool(np.nanmax(np.abs(ow)) > 0) if ow.size else False
        extra = ""
        # ---- a checkpoint in the middle of the run: a second ctx that loads it continues like the first, bit for bit (weights, traces / fa_td / backups /
        # lists / epsilons, step counter travel in the file; states, actions, the episodes' step counts and -- register-family loops -- the carried
        # Q(s,.) through the setters of ABI 8)
        if not bad and rng.random() < 0.5:
            import tempfile
            more = int(rng.choice([5, 23]))
            with tempfile.TemporaryDirectory() as td, ra.Context(**dev) as c2:
                path = os.path.join(td, "w.rsrlw")
                c.save_weights(path)
                st0, ac0, ep0, qc0 = c.states.copy(), c.actions.copy(), c.episode_steps, c.q_carry
                c2.reset()
                c2.train(3, want_stats=False)                      # (something to overwrite)
                c2.load_weights(path)
                c2.states, c2.actions, c2.episode_steps = st0, ac0, ep0
                if qc0 is not None:
                    c2.q_carry = qc0
                c.train(more, want_stats=False)
                c2.train(more, want_stats=False)
                getattr(run, method)(more, **mkw)                  # (the oracle keeps pace: later legs compare against it again)
                same = np.array_equal(c.states, c2.states, equal_nan=True) and np.array_equal(c.actions, c2.actions) and c.step_count == c2.step_count
                for i in ([0] if shared else sorted(set([0, n - 1]))):
                    same = same and np.array_equal(c.get_weights(i), c2.get_weights(i), equal_nan=True)
                    if family == "sparse_lambda" or dev["algo"] in LAMBDA + (ra.TD_LAMBDA,):
                        same = same and np.array_equal(c.get_traces(n - 1), c2.get_traces(n - 1), equal_nan=True)
                    if dev["algo"] == ra.GREEDY_GQ:
                        same = same and np.array_equal(c.get_td_weights(i), c2.get_td_weights(i), equal_nan=True)
                if not same:
                    bad.append("checkpoint resume")
                extra += " +ckpt"
        # ---- Domain::rollout under the greedy policy from the learned weights (lib.rs:334-409): episode lengths as the oracle's, learner for learner
        if not bad and not shared and finite and dev["algo"] not in PRED and family in ("reg", "tile", "generic") and rng.random() < 0.2:
            lim = int(rng.choice([30, 120]))
            n_d, _ = c.rollout_greedy(lim)
            n_o, _ = run.rollout_greedy(lim)
            if not np.array_equal(n_d, n_o):
                bad.append(f"rollout_greedy ({int((n_d != n_o).sum())} of {n} learners)")
            extra += " +rollout"
            # ... and under any of the four policies (lib.rs:448-479 takes any closure), twice: the second call draws from the next stream
            pol = int(rng.integers(0, 4))
            for call in range(2):
                rd = c.rollout_policy(pol, lim, epsilon=0.25, tau=0.7)
                n_o, _, a_o = run.rollout_policy(pol, lim, epsilon=0.25, tau=0.7, call=call)
                if not (np.array_equal(rd["n_states"], n_o) and np.array_equal(rd["actions"], a_o)):
                    bad.append(f"rollout_policy {pol} call {call} ({int((rd['n_states'] != n_o).sum())} of {n} learners)")
                    break
        # ---- set_weights / get_weights: every layout (rows of learners, tile tables, the wave family's lane order, bf16 storage) gives back what went in
        if not bad and rng.random() < 0.3:
            i = int(rng.integers(0, 1 if shared else n))
            w_in = (rng.normal(size=c.get_weights(0).shape) * 0.3).astype(np.float32)
            if dev.get("weight_dtype", ra.W_F32) == ra.W_BF16:
                w_in = (w_in.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)      # bf16-representable: stored exactly
            keep = c.get_weights(i).copy()
            other = c.get_weights((i + 1) % n).copy() if (not shared and n > 1) else None
            c.set_weights(w_in, i)
            if not np.array_equal(c.get_weights(i), w_in):
                bad.append("set/get weights")
            if other is not None and not np.array_equal(c.get_weights((i + 1) % n), other, equal_nan=True):
                bad.append("set_weights touched a neighbour")
            c.set_weights(keep, i)
            extra += " +setget"
        # ---- Domain::default() + the first action in the middle of a run: reset() on both sides, then on
        if not bad and family != "sparse_lambda" and rng.random() < 0.25:
            more = int(rng.choice([4, 19]))
            (run.reset_wave if method == "train_wave" else run.reset)()
            c.reset()
            getattr(run, method)(more, **mkw)
            c.train(more, want_stats=False)
            if not (np.array_equal(c.states.T, run.state, equal_nan=True) and np.array_equal(c.actions, run.action)):
                bad.append("after reset: states / actions")
            ow = run.weights
            w0 = c.get_weights(0)
            if not np.array_equal(w0, (ow if shared else ow[0]).reshape(w0.shape), equal_nan=True):
                bad.append("after reset: weights")
            extra += " +reset"
        # ---- sharding by global env id: two ctxs with env offsets reproduce the one (per-learner weights: no communication)
        if not bad and not shared and n >= 3 and rng.random() < 0.3:
            n1 = int(rng.integers(1, n))
            parts = []
            for off, cnt in ((0, n1), (n1, n - n1)):
                with ra.Context(**dict(dev, n_envs=cnt, env_offset=dev["env_offset"] + off)) as cs:
                    cs.reset()
                    cs.train(17, want_stats=False)
                    parts.append((cs.states.copy(), cs.actions.copy(), cs.get_weights(0).copy(), cs.get_weights(cnt - 1).copy()))
            with ra.Context(**dev) as cf:
                cf.reset()
                cf.train(17, want_stats=False)
                okp = np.array_equal(np.concatenate([parts[0][0], parts[1][0]], axis=1), cf.states, equal_nan=True) and \
                    np.array_equal(np.concatenate([parts[0][1], parts[1][1]]), cf.actions) and \
                    np.array_equal(parts[0][2], cf.get_weights(0), equal_nan=True) and np.array_equal(parts[0][3], cf.get_weights(n1 - 1), equal_nan=True) and \
                    np.array_equal(parts[1][2], cf.get_weights(n1), equal_nan=True) and np.array_equal(parts[1][3], cf.get_weights(n - 1), equal_nan=True)
            if not okp:
                bad.append("sharded != unsharded")
            extra += " +shard"
        # ---- Handler::handle on caller-supplied transitions, where the oracle's handle_* restates the kernel (the reference-order families)
        if not bad and not shared and finite and family in ("tile", "generic") and dev["algo"] in LAMBDA + PRED + (ra.GREEDY_GQ,) + ONE_STEP[:3] \
                and dev.get("agent_policy") is None and "epsilon_decay" not in dev and rng.random() < 0.5:     # (orc.handle_* take the agent's epsilon, not a learner's)
            m = min(n, 4)
            a0 = c.actions
            frm, nxt, rew, term = c.domain_step(a0)
            t_h = c.step_count
            W0 = [c.get_weights(i).copy() for i in range(m)]
            aux = None
            if dev["algo"] in LAMBDA + (ra.TD_LAMBDA,):
                aux = [c.get_traces(i).copy() for i in range(m)]
            elif dev["algo"] == ra.GREEDY_GQ:
                aux = [c.get_td_weights(i).copy() for i in range(m)]
            tdv = c.handle(frm, a0, rew, nxt, term)
            for i in range(m):
                W = W0[i].copy()
                x = orc.draw(dev["seed"], dev["env_offset"] + i, t_h, orc.BLK_INNER)
                if dev["algo"] in LAMBDA:
                    Z = aux[i].copy()
                    d = orc.handle_lambda(ag, W, Z, frm[:, i], a0[i], rew[i], nxt[:, i], term[i], x, "f32d")
                    okh = np.array_equal(c.get_traces(i), Z, equal_nan=True)
                elif dev["algo"] in PRED:
                    Z = aux[i].copy() if aux else None
                    d = orc.handle_td(ag, W, Z, frm[:, i], rew[i], nxt[:, i], term[i], "f32d")
                    okh = aux is None or np.array_equal(c.get_traces(i), Z, equal_nan=True)
                elif dev["algo"] == ra.GREEDY_GQ:
                    V = aux[i].copy()
                    d = orc.handle_gq(ag, W, V, frm[:, i], a0[i], rew[i], nxt[:, i], term[i], "f32d")
                    okh = np.array_equal(c.get_td_weights(i), V, equal_nan=True)
                else:
                    d = orc.handle(ag, W, frm[:, i], a0[i], rew[i], nxt[:, i], term[i], x, "f32d")
                    okh = True
                if not (okh and np.array_equal(c.get_weights(i), W, equal_nan=True) and (np.float32(d) == tdv[i] or (np.isnan(d) and np.isnan(tdv[i])))):
                    bad.append(f"handle[{i}]")
            extra += " +handle"
    if bad:
        return "MISMATCH", tag + f"  MISMATCH {bad}", dev
    return "ok", tag + f"  ok{extra}{'' if finite else ' (non-finite, NaN for NaN)'}{'' if moved else ' (weights did not move)'}", dev


def run_big(rng, idx):
    """The kernels only large learner counts select (four lanes per learner from 131 072 learners, 128-byte line stores beyond the Infinity Cache, two
    waves per SIMD in the fused loop): per-learner weights make every learner independent and the draws are keyed by the GLOBAL env id, so the oracle
    replays SLICES of the batch (env_offset) and each must match the full-size device run bit for bit."""
    domain = int(rng.integers(0, 3))
    kw = dict(domain=domain, order=int(rng.integers(1, 6)) if domain == 0 else 1, algo=int(rng.choice(ONE_STEP)), policy=int(rng.choice([0, 1, 1, 2])),
              epsilon=0.2, tau=1.0, gamma=float(rng.choice([0.9, 0.99])), alpha=float(rng.choice([0.5, 1.0])), seed=int(rng.integers(0, 1 << 20)),
              max_episode_steps=int(rng.choice([0, 7, 200])))
    feats = (kw["order"] + 1) ** N_DIM[domain]
    kw["lr"] = 0.1 / feats
    n = int(rng.choice([131072, 131072 + 77, 200003, 262144, 700003]))
    spl = int(rng.choice([1, 1, 0, 5]))
    K = int(rng.choice([5, 12]))
    tag = f"{idx:4d} big           {NAMES[kw['algo']]:13s} dom {domain} N {n:6d} K {K:3d} spl {spl}"
    with ra.Context(n_envs=n, steps_per_launch=spl, **kw) as c:
        c.reset()
        c.train(K // 2, want_stats=False)
        c.train(K - K // 2, want_stats=False)
        S, Aact = c.states, c.actions
        bad = []
        for off in sorted(set([0, int(rng.integers(0, n - 70)), n - 64])):
            ag = orc.make_agent(env_offset=off, **kw)
            run = orc.Run(ag, 64, "f32d")
            run.reset()
            run.train_dev(K // 2)
            run.train_dev(K - K // 2)
            if not (np.array_equal(S[:, off:off + 64].T, run.state) and np.array_equal(Aact[off:off + 64], run.action)):
                bad.append(f"states / actions of learners {off}..")
            for j in (0, 63):
                if not np.array_equal(c.get_weights(off + j), run.weights[j]):
                    bad.append(f"weights[{off + j}]")
    return ("MISMATCH" if bad else "ok"), tag + (f"  MISMATCH {bad}" if bad else "  ok"), dict(kw, n_envs=n, steps_per_launch=spl)


def main():
    if os.environ.get("FUZZ_BIG"):
        n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
        rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        counts, failures = {}, []
        for idx in range(n_cases):
            status, line, dev = run_big(rng, idx)
            counts[status] = counts.get(status, 0) + 1
            print(line, flush=True)
            if status != "ok":
                failures.append({"case": idx, "line": line, "config": dev})
        print("SUMMARY " + json.dumps({"cases": n_cases, "counts": counts, "failures": failures}, default=str), flush=True)
        sys.exit(1 if failures else 0)
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    counts, failures = {}, []
    for idx in range(n_cases):
        try:
            status, line, dev = run_case(rng, idx)
        except Exception as e:      # noqa: BLE001
            status, line, dev = "ERROR", f"{idx:4d} ERROR {type(e).__name__}: {str(e)[:200]}", None
        counts[status] = counts.get(status, 0) + 1
        print(line, flush=True)
        if status in ("MISMATCH", "ERROR"):
            failures.append({"case": idx, "line": line, "config": dev})
    print("SUMMARY " + json.dumps({"cases": n_cases, "seed": seed, "counts": counts, "failures": failures}, default=str), flush=True)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
