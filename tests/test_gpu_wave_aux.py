"""GreedyGQ, TD and TDLambda on the order-7 WAVE family (F = 4096, one wavefront per learner; rsrl_amd/csrc/kernels_wave_aux.hpp) -- the reference's
agents are generic over the approximator (greedy_gq.rs:49-60, prediction/td/td.rs:25-32, td_lambda.rs:25-40).  Bitwise against the oracle's
wave-order loop (f32d), single transitions against the f64 oracle, V(s) / trait-granular entry points, checkpoint."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


def rand_states(orc, domain, M, seed, shrink=0.5):
    lo, hi = orc.domain_bounds(domain)
    rng = np.random.default_rng(seed)
    mid, half = (lo + hi) / 2, (hi - lo) / 2 * shrink
    return (mid[:, None] + half[:, None] * (2 * rng.random((len(lo), M)) - 1)).astype(np.float32)


# (name, domain, device kwargs, oracle kwargs, steps).  TDLambda's weight step is the TD error itself -- ScaledGradientUpdate{alpha: td_error}, no
# learning rate (td_lambda.rs:59-62): on 4096 features it diverges within a few updates in ANY arithmetic (the f64 reference included), so those
# runs stop while every value is still finite (the bits are compared all the same).
CASES = [
    ("gq_cartpole_egreedy", 1, dict(algo=6, policy=1, epsilon=0.2, lr=0.0002, lr_td=0.001, gamma=0.99), 60),
    ("gq_acrobot_softmax", 2, dict(algo=6, policy=2, tau=0.7, lr=0.0001, lr_td=0.0005, gamma=0.95), 40),
    ("gq_acrobot_greedy", 2, dict(algo=6, policy=0, lr=0.0002, lr_td=0.002, gamma=0.99), 40),
    ("td_cartpole", 1, dict(algo=7, policy=3, lr=0.0002, gamma=0.99), 60),
    ("td_acrobot", 2, dict(algo=7, policy=3, lr=0.0001, gamma=0.9), 40),
    ("tdl_cartpole_accumulate", 1, dict(algo=8, policy=3, gamma=0.99, lam=0.8, trace=0, alpha=0.1), 16),
    ("tdl_cartpole_saturate", 1, dict(algo=8, policy=3, gamma=0.99, lam=0.8, trace=1, alpha=0.1), 16),
    ("tdl_acrobot_dutch", 2, dict(algo=8, policy=3, gamma=0.99, lam=0.7, trace=2, alpha=0.2), 12),
    # QSigma (q_sigma.rs:80-202): the n-step backup ring in memory, one wave per learner
    ("qsigma_cartpole_n3_half", 1, dict(algo=9, policy=1, epsilon=0.2, lr=0.0002, alpha=0.5, gamma=0.99, sigma=0.5, n_steps=3), 60),
    ("qsigma_acrobot_n1_sarsa", 2, dict(algo=9, policy=1, epsilon=0.1, lr=0.0001, alpha=1.0, gamma=0.95, sigma=1.0, n_steps=1), 40),
    ("qsigma_cartpole_n4", 1, dict(algo=9, policy=1, epsilon=0.3, lr=0.0002, alpha=0.5, gamma=0.99, sigma=0.3, n_steps=4), 60),      # (sigma = 0 with n > 1 learns nothing from a terminal-only reward: the terminal entry has pi = 0, q_sigma.rs:142-153)
]


def _okw(kw):
    o = dict(kw)
    return o


BF16_CASES = [(n + "_bf16", d, dict(k, weight_dtype=1), K) for n, d, k, K in CASES if n not in ("gq_acrobot_greedy", "td_acrobot", "tdl_cartpole_accumulate", "qsigma_cartpole_n4")]


@pytest.mark.parametrize("name,domain,kw,K", CASES + BF16_CASES, ids=[c[0] for c in CASES + BF16_CASES])
def test_free_running_bitwise_vs_wave_order_oracle(ra, orc, name, domain, kw, K):
    # *_bf16 (round 6): W stored as bf16 with stochastic rounding of every stored entry (Philox block 16 + 64 * column + lane), fa_td's weights / the trace f32
    bf16 = kw.get("weight_dtype", 0) == 1
    kw = {k: v for k, v in kw.items() if k != "weight_dtype"}
    N = 7                                                      # two thread blocks, the second one partially filled
    ag = orc.make_agent(domain=domain, order=7, seed=9, max_episode_steps=13, **_okw(kw))
    run = orc.Run(ag, N, "f32d")
    if kw["algo"] in (6, 9):
        run.reset_wave()
    else:
        run.reset()
    run.train_wave(K, bf16=bf16)
    pred = kw["algo"] in (7, 8)
    for spl in (0, 1, 5):                                      # any split into launches: one, K, ceil(K / 5)
        with ra.Context(domain=domain, order=7, n_envs=N, seed=9, max_episode_steps=13, steps_per_launch=spl, weight_dtype=ra.W_BF16 if bf16 else ra.W_F32, **kw) as c:
            assert c.F == 4096 and c.n_out == (1 if pred else c.A)
            c.reset()
            st = c.train(K)
            assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action), (name, spl)
            for i in range(N):
                assert np.array_equal(c.get_weights(i), run.weights[i]), (name, spl, i)
            if kw["algo"] == 6:
                for i in (0, N - 1):
                    assert np.array_equal(c.get_td_weights(i), run.traces[i]), (name, spl, i)
            if kw["algo"] == 8:
                for i in (0, N - 1):
                    assert np.array_equal(c.get_traces(i), run.traces[i]), (name, spl, i)
            assert st["env_steps"] == N * K
            assert np.isfinite(run.weights).all() and np.abs(run.weights).max() > 0
            if bf16:
                assert np.all((run.weights.view(np.uint32) & 0xffff) == 0)


def test_qsigma_single_transitions_vs_f64(ra, orc):
    # three handle calls per learner through a 2-step backup: the second and third update the anchor
    M, n = 5, 2
    rng = np.random.default_rng(5)
    kw = dict(gamma=0.97, lr=0.001, alpha=0.5, sigma=0.5, n_steps=n, epsilon=0.2)
    ag = orc.make_agent(domain=1, order=7, algo=9, policy=1, seed=4, **kw)
    with ra.Context(domain=1, order=7, algo=9, policy=1, seed=4, n_envs=M, **kw) as c:
        Ws = [(rng.normal(size=(4096, 2)) * 0.02).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(Ws[i], i)
        W64 = [w.astype(np.float64) for w in Ws]
        bks = [orc.QSigmaBackup(n, "f64") for _ in range(M)]
        s = rand_states(orc, 1, M, 33, shrink=0.3)
        c.states = s
        for k in range(3):
            a = rng.integers(0, 2, M).astype(np.int32)
            frm, nxt, rew, term = c.domain_step(a)
            td = c.handle(frm, a, rew, nxt, term)
            for i in range(M):
                d = bks[i].handle(ag, W64[i], frm[:, i], a[i], rew[i], nxt[:, i], term[i], orc.draw(4, i, k, orc.BLK_INNER))
                assert abs(td[i] - d) <= 1e-4 * (1 + abs(d)), (k, i, td[i], d)
        for i in range(M):
            assert np.max(np.abs(c.get_weights(i) - W64[i])) <= 5e-6
            assert not np.array_equal(c.get_weights(i), Ws[i])      # the anchor did move


@pytest.mark.parametrize("algo,domain,bf16", [(6, 1, False), (6, 2, False), (7, 2, False), (8, 1, False), (6, 2, True), (7, 1, True), (8, 2, True)])
def test_single_transitions_vs_f64(ra, orc, algo, domain, bf16):
    # Handler::handle on caller-supplied transitions, against the reference-precision oracle (identical fp32-representable inputs).
    # bf16 (round 6): from bf16-representable weights, every stored entry of W lands within ONE bf16 ulp of the f64 result (stochastic rounding moves a value to
    # one of its two neighbours) and the roundings are unbiased over the entries; fa_td's weights / the trace stay f32
    M = 6
    rng = np.random.default_rng(algo * 10 + domain)
    pol = 3 if algo in (7, 8) else 1
    kw = dict(gamma=0.97, lr=0.001, lr_td=0.002, lam=0.6, trace=0, alpha=0.3)
    ag = orc.make_agent(domain=domain, order=7, algo=algo, policy=pol, seed=4, **kw)
    A = 2 if domain == 1 else 3
    n_out = 1 if algo in (7, 8) else A
    s = rand_states(orc, domain, M, 21)
    a = rng.integers(0, A, M).astype(np.int32)
    with ra.Context(domain=domain, order=7, algo=algo, policy=pol, seed=4, n_envs=M, weight_dtype=ra.W_BF16 if bf16 else ra.W_F32, **kw) as c:
        Ws = [(rng.normal(size=(4096, n_out)) * 0.02).astype(np.float32) for _ in range(M)]
        if bf16:
            Ws = [(w.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32) for w in Ws]
        Xs = [(rng.normal(size=(4096, n_out)) * 0.02).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(Ws[i], i)
            if algo == 6:
                c.set_td_weights(Xs[i], i)
            if algo == 8:
                c.set_traces(Xs[i], i)
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        term[0] = 1                                            # one terminal transition in the batch
        if algo in (7, 8):
            v = c.q_evaluate(frm)                              # V(s): one value per state
            assert v.shape == (1, M)
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            W, X = Ws[i].astype(np.float64), Xs[i].astype(np.float64)
            if algo == 6:
                d = orc.handle_gq(ag, W, X, frm[:, i], a[i], rew[i], nxt[:, i], term[i], "f64")
                assert np.max(np.abs(c.get_td_weights(i) - X)) <= 3e-6 * (1 + abs(d))
            else:
                w1, z1 = W.reshape(-1).copy(), (X.reshape(-1).copy() if algo == 8 else None)
                assert abs(v[0, i] - orc.v_evaluate(ag, w1, frm[:, i], "f64")) <= 5e-5 * (1 + abs(v[0, i]))
                d = orc.handle_td(ag, w1, z1, frm[:, i], rew[i], nxt[:, i], term[i], "f64")
                W = w1.reshape(4096, 1)
                if algo == 8:
                    assert np.max(np.abs(c.get_traces(i).reshape(-1) - z1)) <= 3e-6
            assert abs(td[i] - d) <= 1e-4 * (1 + abs(d)), (i, td[i], d)
            tol = 3e-6 * (1 + abs(d)) if algo != 8 else 1e-5 * (1 + abs(d)) * (1 + np.abs(Xs[i]).max() * 50)
            if not bf16:
                assert np.max(np.abs(c.get_weights(i) - W)) <= tol, (i, np.max(np.abs(c.get_weights(i) - W)), tol)
            else:
                Wd = c.get_weights(i)
                assert np.all((Wd.view(np.uint32) & 0xffff) == 0)
                ulp = 2.0 ** (np.floor(np.log2(np.abs(W) + 1e-300)) - 7)
                err = (Wd.astype(np.float64) - W) / ulp
                moved = np.abs(W - Ws[i].astype(np.float64)) > 0                       # the entries the step stored
                assert moved.any() and np.max(np.abs(err[moved])) <= 1.0 + 1e-3 + tol / ulp[moved].min(), (i, np.max(np.abs(err[moved])))
                assert np.array_equal(Wd[~moved], Ws[i][~moved])
                assert abs(err[moved].mean()) <= 0.06, (i, err[moved].mean())               # unbiased: the mean of ~4 096+ roundings uniform in (-1, 1) ulp


def test_wave_aux_entry_points_and_checkpoint(ra, tmp_path):
    # GreedyGQ on the wave family behind the whole trait surface: policy ops, rollouts, checkpoint with the second matrix
    kw = dict(domain=1, order=7, algo=6, policy=1, epsilon=0.1, lr=0.0002, lr_td=0.001, gamma=0.99, n_envs=5, seed=2, max_episode_steps=0)     # (no step cap: the per-learner episode counters are not part of a checkpoint)
    with ra.Context(**kw) as c, ra.Context(**kw) as d:
        c.reset()
        c.train(50)
        s = c.states
        assert c.q_evaluate(s).shape == (2, 5) and c.policy_mode(s).shape == (5,) and c.policy_probs(s).shape == (2, 5)
        n, _ = c.rollout_greedy(40)
        assert n.min() >= 2
        path = str(tmp_path / "gq7.ckpt")
        c.save_weights(path)
        d.load_weights(path)
        for i in range(5):
            assert np.array_equal(c.get_weights(i), d.get_weights(i)) and np.array_equal(c.get_td_weights(i), d.get_td_weights(i))
        d.states, d.actions = c.states, c.actions
        # (the step counter keys the draws: a resumed run needs it too -- it travels in the file)
        c.train(20), d.train(20)
        assert np.array_equal(c.states, d.states) and np.array_equal(c.get_weights(3), d.get_weights(3))
    # prediction agents: no action values
    with ra.Context(domain=2, order=7, algo=7, policy=3, n_envs=3, lr=1e-4) as c:
        c.reset()
        c.train(5)
        with pytest.raises(ra.RsrlHipError):
            c.policy_mode(c.states)
        with pytest.raises(ra.RsrlHipError):
            c.rollout_greedy(10)
        assert c.project(c.states).shape == (4096, 3)
    # QSigma with bf16 weights: the wave family only (round 6)
    with pytest.raises(ra.RsrlHipError):
        ra.Context(domain=0, order=3, algo=9, policy=1, n_envs=2, weight_dtype=ra.W_BF16)
    with ra.Context(domain=2, order=7, algo=9, policy=1, n_envs=2, weight_dtype=ra.W_BF16, n_steps=2, lr=1e-4) as c:
        c.reset(); c.train(8)
        assert np.all((c.get_weights(0).view(np.uint32) & 0xffff) == 0) and np.abs(c.get_weights(0)).max() > 0
    # bf16 (round 6): the granular entry points of GreedyGQ / TD read the bf16 tables
    with ra.Context(domain=2, order=7, algo=6, policy=1, n_envs=3, lr=1e-4, lr_td=1e-4, weight_dtype=ra.W_BF16) as c:
        c.reset(); c.train(6)
        assert c.q_evaluate(c.states).shape == (3, 3) and np.abs(c.get_weights(1)).max() > 0
        assert np.all((c.get_weights(1).view(np.uint32) & 0xffff) == 0)
    with ra.Context(domain=1, order=7, algo=7, policy=3, n_envs=3, lr=1e-4, weight_dtype=ra.W_BF16) as c:
        c.reset(); c.train(6)
        assert np.all(np.isfinite(c.q_evaluate(c.states)))
