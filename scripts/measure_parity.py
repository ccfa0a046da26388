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


# ---------------------------------------------------------------------------------------------------------------------------------
# Every BASELINE.json configuration against the f64 oracle (the reference's precision), at a size the oracle finishes in seconds.
#   teacher-forced: the f64 oracle runs the configuration's driver loop with its successor states rounded to fp32 (oracle.Run.
#     teacher_step) and hands every batch-step's transitions to the device's Handler::handle -- the device and the oracle learn from
#     IDENTICAL fp32-representable inputs for K steps (the k-th handle call draws what the oracle's k-th batch-step drew: SARSA's inner
#     sample, the bf16 rounding bits).  Reported: worst TD error difference along the way, max|dW| and max|dQ| (on the last step's
#     states) at the end, absolute and relative to max(1, max|W|) (SURVEY 8(d)'s form) and to max|W| itself.
#   bf16: the same tape through a bf16-weights ctx and an fp32-weights ctx -> max|W_bf16 - W_f32|, max|W_bf16 - W_f64|, the same for Q,
#     and the model bound sqrt(K) * 2^-8 * max|W| they are stated against (K stochastic roundings of one ulp(bf16) = 2^-7 relative at most,
#     unbiased and independent: a random walk).
#   free-running: device train(K) vs f64 oracle train(K) from the same seed -- trajectories part ways at the first argmax decided by an
#     fp32 rounding, so the comparison is of POPULATION statistics: episodes, sum|delta|, sum of rewards, relative differences.
CONFIGS = {
    "c2": dict(what="configs[1] in small: MountainCar QLearning Fourier(5) eps-greedy, per-env W", M=256, K=1000, K_free=2000, cap=200,
               kw=dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, lr=0.001)),
    "c3": dict(what="configs[2] in small: CartPole SARSA tile coding 8 x 8^4 eps-greedy, ONE shared table", M=512, K=1000, K_free=2000, cap=200,
               kw=dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=1, policy=1, epsilon=0.1, gamma=0.99, lr=0.0125, shared=True)),
    "c4": dict(what="configs[3], one rank in small: MountainCar QLearning Fourier(5) eps-greedy, ONE shared W", M=512, K=1000, K_free=2000, cap=200,
               kw=dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, lr=0.001, shared=True)),
    # configs[4]'s lr = 1e-3 is SURVEY 8(d)'s build-chosen value ("bf16 + lr = 1e-3"): with F = 4096 features |phi|^2 ~ 2048, so one update moves
    # Q(s,a) by lr * |phi|^2 * delta ~ 2.05 delta -- past SGD's stability limit of 2.  The f64 oracle itself diverges under it (mean |delta| per
    # 500 steps: 2.4, 33, 248, 2255), so free-running population statistics mean nothing there; they are taken at lr = 2.5e-4 (c5s: 0.92, 1.0,
    # 1.08, 1.10), and the teacher-forced legs run at both.
    "c5": dict(what="configs[4] in small: Acrobot ExpectedSARSA Fourier(7) Softmax(1.0), per-env W, f32 and bf16 weights, lr 1e-3", M=8, K=200,
               K_free=0, cap=100, bf16=True, kw=dict(domain=2, order=7, algo=2, policy=2, tau=1.0, gamma=0.99, lr=0.001, alpha=1.0)),
    "c5s": dict(what="configs[4] in small at a stable step size: lr 2.5e-4 (lr |phi|^2 ~ 0.5)", M=8, K=200, K_free=2000,
                cap=100, bf16=True, kw=dict(domain=2, order=7, algo=2, policy=2, tau=1.0, gamma=0.99, lr=0.00025, alpha=1.0)),
    # the agents widened onto the order-7 wave family in round 5 (kernels_wave_aux.hpp), teacher-forced against f64 like the configurations above
    # (no bf16 leg for GreedyGQ here: its second update goes to the column argmax of Q(s',.) picks UNDER THE RUN'S OWN W, and near the zero initialisation the
    # action gaps are below bf16's rounding noise -- the bf16 and f32 runs update different columns (measured: |W_bf16 - W_f32| = 0.43 max|W| after 200 steps, 8x the
    # random-walk bound the TD / SARSALambda legs keep: 0.80 / 0.61 of it).  GreedyGQ's bf16 arithmetic is pinned per transition instead:
    # tests/test_gpu_wave_aux.py::test_single_transitions_vs_f64[bf16] -- every stored entry within one bf16 ulp of f64, unbiased.)
    "w7_gq": dict(what="GreedyGQ on Acrobot Fourier(7), eps-greedy: W and fa_td's V", M=6, K=200, K_free=1000, cap=100,
                  kw=dict(domain=2, order=7, algo=6, policy=1, epsilon=0.1, gamma=0.99, lr=0.0002, lr_td=0.001)),
    "w7_td": dict(what="TD (state values) on CartPole Fourier(7), Random behaviour (f32 and bf16 weights)", M=6, K=200, K_free=1000, cap=100, bf16=True,
                  kw=dict(domain=1, order=7, algo=7, policy=3, gamma=0.99, lr=0.0002)),
    "w7_sl": dict(what="SARSALambda (accumulating trace, lambda 0.8) on CartPole Fourier(7), eps-greedy (f32 and bf16 weights, the trace f32)", M=6, K=200, K_free=1000, cap=100,
                  bf16=True, kw=dict(domain=1, order=7, algo=3, policy=1, epsilon=0.1, gamma=0.99, alpha=0.0001, lam=0.8, trace=0)),
}


def _make(cfg, M, seed, cap, weight_dtype=None):
    kw = dict(cfg["kw"])
    shared = kw.pop("shared", False)
    if shared:
        kw["lr"] = kw["lr"] / M                      # the mini-batch rule sums M updates (bench.py scales the same way)
    basis = kw.pop("basis", 0)
    okw = dict(kw, basis=orc.TILE if basis else orc.FOURIER, shared_w=shared, seed=seed, max_episode_steps=cap)
    dkw = dict(kw, basis=ra.TILE_CODING if basis else ra.FOURIER, weight_mode=ra.W_SHARED if shared else ra.W_PER_ENV, seed=seed,
               max_episode_steps=cap, n_envs=M)
    if weight_dtype is not None:
        dkw["weight_dtype"] = weight_dtype
    return orc.make_agent(**okw), dkw, shared


def _weights(c, M, shared):
    return c.get_weights() if shared else np.stack([c.get_weights(i) for i in range(M)])


def teacher_forced(name, M=None, K=None, seed=11):
    cfg = CONFIGS[name]
    M, K = M or cfg["M"], K or cfg["K"]
    ag, dkw, shared = _make(cfg, M, seed, cfg["cap"])
    run = orc.Run(ag, M, "f64")
    run.reset()
    ctxs = {"f32": ra.Context(**dkw)}
    if cfg.get("bf16"):
        ctxs["bf16"] = ra.Context(**dict(dkw, weight_dtype=ra.W_BF16))
    td_err = {k: 0.0 for k in ctxs}
    for _ in range(K):
        t = run.teacher_step()
        frm, to = np.ascontiguousarray(t["frm"].T, dtype=np.float32), np.ascontiguousarray(t["to"].T, dtype=np.float32)
        for k, c in ctxs.items():
            td = c.handle(frm, t["action"], t["reward"].astype(np.float32), to, t["terminal"])
            td_err[k] = max(td_err[k], float(np.max(np.abs(td - t["td"]) / (1 + np.abs(t["td"])))))
    W64 = np.array(run.weights)
    wmax = float(np.abs(W64).max())
    probe = np.ascontiguousarray(t["to"].T, dtype=np.float32)
    if ag.algo in (orc.TD, orc.TD_LAMBDA):
        q64 = np.array([[orc.v_evaluate(ag, W64[i], t["to"][i], "f64") for i in range(M)]])
    else:
        q64 = np.stack([orc.q_evaluate(ag, W64 if shared else W64[i], t["to"][i], "f64") for i in range(M)], axis=1)
    out = {"what": cfg["what"], "learners": M, "steps": K, "max_abs_w_f64": wmax}
    res = {}
    for k, c in ctxs.items():
        Wd = _weights(c, M, shared).astype(np.float64)
        qd = c.q_evaluate(probe).astype(np.float64)
        res[k] = (Wd, qd)
        dw = float(np.abs(Wd - W64).max())
        out[k] = {"td_max_rel": td_err[k], "w_max_abs": dw, "w_rel_to_max1": dw / max(1.0, wmax), "w_rel_to_maxw": dw / max(wmax, 1e-30),
                  "q_max_rel": float(np.max(np.abs(qd - q64) / (1 + np.abs(q64))))}
    if "bf16" in res:
        dwb = float(np.abs(res["bf16"][0] - res["f32"][0]).max())
        out["bf16_vs_f32"] = {"w_max_abs": dwb, "w_rel_to_maxw": dwb / max(wmax, 1e-30),
                              "q_max_abs": float(np.abs(res["bf16"][1] - res["f32"][1]).max()),
                              "model_bound_w": float(np.sqrt(K) * 2.0 ** -8 * wmax),
                              "model": "sqrt(K) * 2^-8 * max|W|: K unbiased roundings of at most one bf16 ulp (2^-7 relative)"}
    for c in ctxs.values():
        c.close()
    run.close()
    return out


def free_running(name, M=None, K=None, seed=11, bf16=False):
    cfg = CONFIGS[name]
    M, K = M or cfg["M"], K or cfg["K_free"]
    ag, dkw, shared = _make(cfg, M, seed, cfg["cap"], ra.W_BF16 if bf16 else None)
    run = orc.Run(ag, M, "f64")
    run.reset()
    so = run.train(K)
    run.close()
    with ra.Context(**dkw) as c:
        c.reset()
        sd = c.train(K)
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
    return {"learners": M, "steps": K, "episodes_device": sd["episodes"], "episodes_f64": so["episodes"],
            "episodes_rel": rel(sd["episodes"], so["episodes"]), "sum_abs_td_rel": rel(sd["sum_abs_td_error"], so["sum_abs_td_error"]),
            "sum_reward_rel": rel(sd["sum_reward"], so["sum_reward"]), "sum_abs_td_f64": so["sum_abs_td_error"]}


def measure_configs(names=("c2", "c3", "c4", "c5", "c5s"), scale=1.0):
    """scale < 1 shortens every leg (bench.py's live sample)"""
    out = {}
    for n in names:
        cfg = CONFIGS[n]
        K, Kf = max(50, int(cfg["K"] * scale)), max(100, int(cfg["K_free"] * scale))
        out[n] = {"teacher_forced": teacher_forced(n, K=K)}
        if cfg["K_free"]:
            out[n]["free_running"] = free_running(n, K=Kf)
            if cfg.get("bf16"):
                out[n]["free_running_bf16"] = free_running(n, K=Kf, bf16=True)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "configs":
        print(json.dumps(measure_configs(scale=float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)))
    else:
        print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)))
