"""GPU parity for GreedyGQ (rsrl/src/control/td/greedy_gq.rs:73-141, rsrl/examples/greedy_gq.rs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


def rand_states(orc, domain, M, seed):
    lo, hi = orc.domain_bounds(domain)
    rng = np.random.default_rng(seed)
    return (lo[:, None] + (hi - lo)[:, None] * rng.random((len(lo), M))).astype(np.float32)


@pytest.mark.parametrize("domain,order", [(0, 3), (0, 5), (1, 1), (2, 1)])
def test_gq_handle(ra, orc, domain, order):
    M = 128
    rng = np.random.default_rng(domain * 7 + order)
    kw = dict(gamma=0.99, lr=0.1, lr_td=0.001)
    ag = orc.make_agent(domain=domain, order=order, algo=orc.GREEDY_GQ, policy=orc.EGREEDY, seed=2, **kw)
    A = ag.n_actions
    s = rand_states(orc, domain, M, 5)
    a = rng.integers(0, A, M).astype(np.int32)
    with ra.Context(domain=domain, order=order, n_envs=M, algo=ra.GREEDY_GQ, policy=1, seed=2, **kw) as c:
        F = c.F
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        term[::5] = 1
        Ws = [(rng.normal(size=(F, A)) * 0.3).astype(np.float32) for _ in range(M)]
        Vs = [(rng.normal(size=(F, A)) * 0.2).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(Ws[i], i)
            c.set_td_weights(Vs[i], i)
        assert np.array_equal(c.get_td_weights(3), Vs[3])
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            W, V = Ws[i].copy(), Vs[i].copy()
            d = orc.handle_gq(ag, W, V, frm[:, i], a[i], rew[i], nxt[:, i], term[i], "f32")
            assert abs(td[i] - d) <= 2e-5 * (1 + abs(d)), (i, td[i], d)
            assert np.max(np.abs(c.get_weights(i) - W)) <= 3e-6 * (1 + abs(d))
            assert np.max(np.abs(c.get_td_weights(i) - V)) <= 3e-6 * (1 + abs(d))


@pytest.mark.parametrize("domain,order,lr", [(0, 3, 0.1), (0, 5, 0.02), (1, 1, 0.05)])
def test_gq_train_vs_oracle_f32(ra, orc, domain, order, lr):
    # examples/greedy_gq.rs: Fourier(3), fa_q SGD(0.1), fa_td SGD(0.001), eps-greedy(0.1), gamma 0.99, 1000-step episodes
    N, K = 128, 100
    kw = dict(gamma=0.99, lr=lr, lr_td=0.001, epsilon=0.1)
    ag = orc.make_agent(domain=domain, order=order, algo=orc.GREEDY_GQ, policy=orc.EGREEDY, seed=9, max_episode_steps=40, **kw)
    run = orc.Run(ag, N, "f32")
    run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, order=order, n_envs=N, algo=ra.GREEDY_GQ, policy=1, seed=9, max_episode_steps=40, **kw) as c:
        c.reset()
        st = c.train(K)
        same = np.all(np.abs(c.states.T - run.state) <= 1e-5 * (1 + np.abs(run.state)), axis=1) & (c.actions == run.action)
        assert same.mean() >= 0.9, same.mean()
        for i in np.flatnonzero(same)[:16]:
            scale = max(1.0, np.abs(run.weights[i]).max())
            assert np.max(np.abs(c.get_weights(i) - run.weights[i])) <= 2e-5 * scale
            assert np.max(np.abs(c.get_td_weights(i) - run.traces[i])) <= 2e-5 * scale
        assert abs(st["episodes"] - ost["episodes"]) <= max(2, 0.05 * ost["episodes"])
        assert abs(st["sum_abs_td_error"] - ost["sum_abs_td_error"]) <= 5e-3 * ost["sum_abs_td_error"]


def test_gq_fused_equals_stepwise_and_resume(ra, tmp_path):
    kw = dict(n_envs=300, order=3, algo=6, policy=1, epsilon=0.1, gamma=0.99, lr=0.05, lr_td=0.001, seed=3, max_episode_steps=60)
    with ra.Context(**kw) as c1, ra.Context(steps_per_launch=1, **kw) as c2:
        c1.reset(); c2.reset()
        c1.train(64); c2.train(64)
        assert np.array_equal(c1.states, c2.states) and np.array_equal(c1.actions, c2.actions)
        assert np.array_equal(c1.get_weights(7), c2.get_weights(7))
        assert np.array_equal(c1.get_td_weights(7), c2.get_td_weights(7))
        with pytest.raises(ra.RsrlHipError):
            c1.get_traces(0)                       # GreedyGQ has no eligibility trace


def test_gq_configurations(ra):
    with ra.Context(n_envs=8, algo=6, basis=ra.TILE_CODING):                # tile coding: built in round 4 (tests/test_gpu_round4.py)
        pass
    with pytest.raises(ra.RsrlHipError):                                    # ... with per-learner tables only
        ra.Context(n_envs=8, algo=6, basis=ra.TILE_CODING, weight_mode=ra.W_SHARED)
    with ra.Context(n_envs=8, algo=6, domain=2, order=7):                   # the order-7 wave family: since round 5 (tests/test_gpu_wave_aux.py) ...
        pass
    with ra.Context(n_envs=8, algo=6, domain=2, order=7, weight_dtype=ra.W_BF16):      # ... and with bf16 weights since round 6 (fa_td's stay f32)
        pass
    with ra.Context(n_envs=8, algo=0) as c:
        with pytest.raises(ra.RsrlHipError):
            c.get_td_weights(0)
