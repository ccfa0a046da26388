"""Committed golden vectors (tests/golden/vectors.json, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them (guards the checker against drift).  GPU: the HIP path against the same
vectors directly, without the live oracle."""
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))


def test_oracle_reproduces_golden(orc):
    o = G["oracle"]
    assert np.allclose(orc.fourier_project(0, 5, [-0.5, 0.0]), o["fourier5_phi_default_state"], rtol=0, atol=1e-15)
    for rec in o["domain_steps"]:
        ns, r, term = orc.domain_step(rec["domain"], np.array(rec["s"], dtype=np.float64), rec["a"])
        assert np.allclose(ns, rec["ns"], rtol=0, atol=1e-14) and r == rec["r"] and term == rec["term"]
    ag = orc.make_agent(domain=1, basis=orc.TILE, n_tilings=8, tiles_per_dim=8)
    for rec in o["cartpole_tile_indices"]:
        assert orc.tile_indices(ag, rec["s"]).tolist() == rec["idx"]
    ref = G["reference"]
    s = orc.domain_reset(orc.CART_POLE)
    for exp in ref["cartpole_steps_action0"]:
        s, _, _ = orc.domain_step(orc.CART_POLE, s, 0)
        assert np.all(np.abs(s - np.array(exp)) < 1e-7)
    assert orc.policy_sample(orc.GREEDY, ref["greedy_argmax"]["q"], (0, 0, 0, 0)) == ref["greedy_argmax"]["a"]
    for case in ref["egreedy_probs"]["cases"]:
        assert np.allclose(orc.policy_probs(orc.EGREEDY, case["q"], eps=ref["egreedy_probs"]["eps"]), case["p"], atol=1e-6)


@pytest.mark.gpu
def test_device_matches_golden():
    import rsrl_amd as ra
    o = G["oracle"]
    # Fourier(5) features of the default state
    with ra.Context(n_envs=1) as c:
        phi = c.project(np.array([[-0.5], [0.0]], dtype=np.float32))[:, 0]
    assert np.max(np.abs(phi - np.array(o["fourier5_phi_default_state"]))) <= 3e-6
    # domain steps (fp32 vs f64: 1e-5; Acrobot 2e-4, see test_gpu_parity_mc)
    for dom in (0, 1, 2):
        recs = [r for r in o["domain_steps"] if r["domain"] == dom]
        S = np.array([r["s"] for r in recs], dtype=np.float32).T
        with ra.Context(domain=dom, order=5 if dom == 0 else 1, n_envs=len(recs)) as c:
            c.states = S
            _, nxt, rew, term = c.domain_step(np.array([r["a"] for r in recs], dtype=np.int32))
        tol = 2e-4 if dom == 2 else 1e-5
        for k, r in enumerate(recs):
            assert np.allclose(nxt[:, k], r["ns"], rtol=tol, atol=tol)
            assert rew[k] == r["r"] and bool(term[k]) == r["term"]
    # tile indices: bit-exact
    recs = o["cartpole_tile_indices"]
    with ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, n_envs=len(recs), weight_mode=ra.W_SHARED) as c:
        idx = c.tile_indices(np.array([r["s"] for r in recs], dtype=np.float32).T)
    for k, r in enumerate(recs):
        assert idx[:, k].tolist() == r["idx"]
    # one update per agent from a fixed non-zero W
    W0 = np.array(o["W0"], dtype=np.float32)
    for u in o["updates"]:
        with ra.Context(n_envs=1, algo=u["algo"], policy=u["policy"], seed=5, gamma=u["gamma"], lr=u["lr"], alpha=u["alpha"],
                        epsilon=u["epsilon"], tau=u["tau"]) as c:
            c.set_weights(W0, 0)
            td = c.handle(np.array([u["s"]], dtype=np.float32).T, np.array([u["a"]], dtype=np.int32),
                          np.array([u["r"]], dtype=np.float32), np.array([u["ns"]], dtype=np.float32).T,
                          np.array([u["term"]], dtype=np.uint8))
            W = c.get_weights(0)
        assert abs(td[0] - u["delta"]) <= 2e-5 * (1 + abs(u["delta"]))
        assert np.max(np.abs(W[:, 2] - np.array(u["W_col2_after"]))) <= 1e-6 * (1 + abs(u["delta"]))
        assert np.array_equal(W[:, :2], W0[:, :2])
    # 1000 teacher-forced updates
    tf = o["teacher_forced"]
    with ra.Context(n_envs=4, **tf["config"]) as c:
        for step in tf["transitions"]:
            c.handle(np.array([e["s"] for e in step], dtype=np.float32).T, np.array([e["a"] for e in step], dtype=np.int32),
                     np.array([e["r"] for e in step], dtype=np.float32), np.array([e["ns"] for e in step], dtype=np.float32).T,
                     np.array([e["term"] for e in step], dtype=np.uint8))
        Wd = np.stack([c.get_weights(i) for i in range(4)])
    Wo = np.array(tf["W_after"])
    assert np.max(np.abs(Wd - Wo)) <= 1e-3 * max(1.0, np.max(np.abs(Wo)))
    # greedy rollout from fixed weights
    gr = o["greedy_rollout"]
    W = np.array(gr["W"], dtype=np.float32)
    with ra.Context(n_envs=len(W), policy=1) as c:
        for i in range(len(W)):
            c.set_weights(W[i], i)
        n, tot = c.rollout_greedy(gr["limit"])
    agree = n == np.array(gr["n_states"])
    assert agree.mean() >= 0.85
    assert np.all(tot[agree] == np.array(gr["total_reward"])[agree])


# ---- round 2: device-order (bitwise) vectors and QSigma, tests/golden/vectors_r02.json (made by make_golden_r02.py) ----
G2 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors_r02.json")))["oracle"]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _digest(a):
    b = _bits(a).astype(np.uint64).ravel()
    return int((b * (2 * np.arange(b.size, dtype=np.uint64) + 1)).sum(dtype=np.uint64))


def test_oracle_reproduces_golden_r02(orc):
    g = G2["c2_device_order"]
    run = orc.Run(orc.make_agent(**g["config"]), g["n_envs"], "f32d")
    run.reset()
    st = run.train_dev(g["steps"])
    assert _bits(run.state).tolist() == g["states"] and run.action.tolist() == g["actions"]
    assert _bits(run.weights[0]).tolist() == g["w_learner0"] and _digest(run.weights) == g["w_digest_all"] and st["episodes"] == g["episodes"]
    g = G2["c5_wave_order_bf16"]
    run = orc.Run(orc.make_agent(**g["config"]), g["n_envs"], "f32d")
    run.reset_wave()
    run.train_wave(g["steps"], bf16=True)
    assert _bits(run.state).tolist() == g["states"] and run.action.tolist() == g["actions"]
    assert [_digest(run.weights[i]) for i in range(g["n_envs"])] == g["w_digest"]
    g = G2["qsigma"]
    ag = orc.make_agent(**g["config"])
    W = np.array(g["W0"], dtype=np.float64)
    bk = [orc.QSigmaBackup(g["config"]["n_steps"], "f64") for _ in range(2)]
    for k, stp in enumerate(g["steps"]):
        for i in range(2):
            d = bk[i].handle(ag, W[i], np.array(stp["s"])[:, i], stp["a"][i], -1.0, np.array(stp["ns"])[:, i], stp["term"][i], orc.draw(4, i, k, orc.BLK_INNER))
            assert abs(d - stp["residual_f64"][i]) <= 1e-13
    assert np.max(np.abs(W - np.array(g["W_after_f64"]))) <= 1e-13


@pytest.mark.gpu
def test_device_matches_golden_r02_bitwise():
    # the HIP path against the committed device-order vectors, without the live oracle: bit for bit
    import rsrl_amd as ra
    g = G2["c2_device_order"]
    with ra.Context(n_envs=g["n_envs"], **g["config"]) as c:
        c.reset()
        st = c.train(g["steps"])
        assert _bits(c.states.T).tolist() == g["states"] and c.actions.tolist() == g["actions"]
        assert _bits(c.get_weights(0)).tolist() == g["w_learner0"] and st["episodes"] == g["episodes"]
        assert _digest(np.stack([c.get_weights(i) for i in range(g["n_envs"])])) == g["w_digest_all"]
    g = G2["c5_wave_order_bf16"]
    with ra.Context(n_envs=g["n_envs"], weight_dtype=ra.W_BF16, **g["config"]) as c:
        c.reset()
        c.train(g["steps"])
        assert _bits(c.states.T).tolist() == g["states"] and c.actions.tolist() == g["actions"]
        assert [_digest(c.get_weights(i)) for i in range(g["n_envs"])] == g["w_digest"]
        assert _bits(c.get_weights(0)[:4]).tolist() == g["w0_first_rows"]
    g = G2["qsigma"]
    W0 = np.array(g["W0"], dtype=np.float32)
    with ra.Context(n_envs=2, **g["config"]) as c:
        for i in range(2):
            c.set_weights(W0[i], i)
        for stp in g["steps"]:
            td = c.handle(np.array(stp["s"], dtype=np.float32), np.array(stp["a"], dtype=np.int32), -np.ones(2, np.float32),
                          np.array(stp["ns"], dtype=np.float32), np.array(stp["term"], dtype=np.uint8))
            assert _bits(td).tolist() == stp["residual_f32d_bits"]
            assert np.max(np.abs(td - np.array(stp["residual_f64"]))) <= 5e-5 * (1 + np.max(np.abs(stp["residual_f64"])))
        W = np.stack([c.get_weights(i) for i in range(2)])
        assert _bits(W).tolist() == g["W_after_f32d_bits"]
        assert np.max(np.abs(W - np.array(g["W_after_f64"]))) <= 2e-5 * (1 + np.abs(np.array(g["W_after_f64"])).max())


# ---- round 3: trace / prediction agents on tile coding and on the wave family, tests/golden/vectors_r03.json (make_golden_r03.py) ----
G3 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors_r03.json")))["oracle"]


def _check_r03(g, states_t, actions, weights, traces, st):
    n = g["n_envs"]
    assert _bits(states_t).tolist() == g["states"] and list(actions) == g["actions"]
    assert st["episodes"] == g["episodes"] and st["episodes_truncated"] == g["episodes_truncated"]
    assert [_digest(weights(i)) for i in range(n)] == g["w_digest"]
    assert [_digest(traces(i)) for i in range(n)] == g["z_digest"]


def test_oracle_reproduces_golden_r03(orc):
    for key, wave in (("sarsa_lambda_tiles", False), ("td_lambda_tiles", False), ("q_lambda_wave", True)):
        g = G3[key]
        run = orc.Run(orc.make_agent(**g["config"]), g["n_envs"], "f32d")
        if wave:
            run.reset_wave(); st = run.train_wave(g["steps"])
        else:
            run.reset(); st = run.train(g["steps"])
        _check_r03(g, run.state, run.action.tolist(), lambda i: run.weights[i], lambda i: run.traces[i], st)


@pytest.mark.gpu
def test_device_matches_golden_r03_bitwise():
    # the HIP path against the committed vectors, without the live oracle: states, actions, weights and traces bit for bit
    import rsrl_amd as ra
    for key in ("sarsa_lambda_tiles", "td_lambda_tiles", "q_lambda_wave"):
        g = G3[key]
        with ra.Context(n_envs=g["n_envs"], **g["config"]) as c:
            c.reset()
            a, b = g["steps"] // 3, g["steps"] - g["steps"] // 3
            s1, s2 = c.train(a), c.train(b)                    # two launches: the boundary is invisible
            st = {k: s1[k] + s2[k] for k in ("episodes", "episodes_truncated")}
            _check_r03(g, c.states.T, c.actions.tolist(), c.get_weights, c.get_traces, st)


# ---- round 4: the per-learner epsilon schedule, rollouts under a sampling policy, GreedyGQ / QSigma / TD off the register family:
# tests/golden/vectors_r04.json (make_golden_r04.py) ----
G4 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors_r04.json")))["oracle"]
R4_TRAIN = ("sarsa_lambda_eps_schedule", "qlearning_eps_schedule", "greedy_gq_tiles", "q_sigma_generic_fourier", "td_generic_fourier")


def _check_r04(g, states_t, actions, weights, aux, eps, st):
    n = g["n_envs"]
    assert _bits(states_t).tolist() == g["states"] and list(actions) == g["actions"]
    assert st["episodes"] == g["episodes"] > 0 and st["episodes_truncated"] == g["episodes_truncated"]
    assert [_digest(weights(i)) for i in range(n)] == g["w_digest"]
    if g["aux"]:
        assert [_digest(aux(i)) for i in range(n)] == g["z_digest"]
    if "eps_bits" in g:
        assert _bits(eps()).tolist() == g["eps_bits"]
        if g["config"]["domain"] == 1:
            assert len(set(g["eps_bits"])) > 1                   # CartPole: the learners' episodes end at different steps -> different epsilons


def _check_rollouts(g, call_fn):
    for k, rec in enumerate(g["calls"]):
        n_states, total, acts = call_fn(rec["policy"], g["limit"], rec["kw"], k)
        assert n_states.tolist() == rec["n_states"] and _bits(total).tolist() == rec["total_reward_bits"]
        ref = np.array(rec["actions"])
        for i, ns in enumerate(rec["n_states"]):                 # the actions actually taken: ns - 1 of them
            assert acts[:ns - 1, i].tolist() == ref[:ns - 1, i].tolist(), (k, i)


def test_oracle_reproduces_golden_r04(orc):
    for key in R4_TRAIN:
        g = G4[key]
        run = orc.Run(orc.make_agent(**g["config"]), g["n_envs"], "f32d"); run.reset()
        st = (run.train_dev if g["loop"] == "dev" else run.train)(g["steps"])
        _check_r04(g, run.state, run.action.tolist(), lambda i: run.weights[i], lambda i: run.traces[i], lambda: run.eps, st)
    g = G4["rollout_policy"]
    run = orc.Run(orc.make_agent(**g["config"]), g["n_envs"], "f32d"); run.reset(); run.train_dev(g["steps"])
    _check_rollouts(g, lambda policy, limit, kw, k: run.rollout_policy(policy, limit, call=k, **kw))


@pytest.mark.gpu
def test_device_matches_golden_r04_bitwise():
    # the HIP path against the committed vectors, without the live oracle
    import rsrl_amd as ra
    for key in R4_TRAIN:
        g = G4[key]
        with ra.Context(n_envs=g["n_envs"], **g["config"]) as c:
            c.reset()
            a, b = g["steps"] // 3, g["steps"] - g["steps"] // 3
            s1, s2 = c.train(a), c.train(b)
            st = {k: s1[k] + s2[k] for k in ("episodes", "episodes_truncated")}
            aux = c.get_td_weights if g["config"]["algo"] == 6 else c.get_traces
            _check_r04(g, c.states.T, c.actions.tolist(), c.get_weights, aux, lambda: c.epsilons, st)
    g = G4["rollout_policy"]
    with ra.Context(n_envs=g["n_envs"], **g["config"]) as c:
        c.reset(); c.train(g["steps"])

        def call(policy, limit, kw, k):
            r = c.rollout_policy(policy, limit, **kw)            # (the ctx numbers its sampling rollouts itself: 0, 1, 2)
            return r["n_states"], r["total_reward"], r["actions"]
        _check_rollouts(g, call)
