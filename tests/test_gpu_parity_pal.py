"""GPU parity for PAL, persistent advantage learning (rsrl/src/control/td/pal.rs:34-60)."""
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


@pytest.mark.parametrize("domain,basis", [(0, "fourier5"), (1, "fourier1"), (1, "tile"), (2, "fourier7")])
def test_pal_handle(ra, orc, domain, basis):
    M = 64 if basis == "fourier7" else 128
    rng = np.random.default_rng(domain + 11)
    kw = dict(gamma=0.97, lr=0.01, alpha=0.4)
    if basis == "tile":
        okw, dkw = dict(basis=orc.TILE, n_tilings=8, tiles_per_dim=8), dict(basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8)
    else:
        o = int(basis[-1])
        okw, dkw = dict(order=o), dict(order=o)
    ag = orc.make_agent(domain=domain, algo=orc.PAL, policy=orc.GREEDY, seed=2, **okw, **kw)
    s = rand_states(orc, domain, M, 5)
    A = ag.n_actions
    a = rng.integers(0, A, M).astype(np.int32)
    with ra.Context(domain=domain, n_envs=M, algo=ra.PAL, policy=0, seed=2, **dkw, **kw) as c:
        F = c.F
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        term[::5] = 1
        Ws = [(rng.normal(size=(F, A)) * (0.3 if F < 100 else 0.02)).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(Ws[i], i)
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            W = Ws[i].copy()
            d = orc.handle(ag, W, frm[:, i], a[i], rew[i], nxt[:, i], term[i], (0, 0, 0, 0), "f32")
            assert abs(td[i] - d) <= 2e-5 * (1 + abs(d)), (i, td[i], d)
            assert np.max(np.abs(c.get_weights(i) - W)) <= 2e-6 * (1 + abs(d))


@pytest.mark.parametrize("domain,order,spl", [(0, 5, 0), (0, 3, 1), (1, 1, 0)])
def test_pal_train_vs_oracle_f32(ra, orc, domain, order, spl):
    N, K = 128, 100
    kw = dict(gamma=0.99, lr=0.005, alpha=0.5, epsilon=0.2)
    ag = orc.make_agent(domain=domain, order=order, algo=orc.PAL, policy=orc.EGREEDY, seed=9, max_episode_steps=40, **kw)
    run = orc.Run(ag, N, "f32")
    run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, order=order, n_envs=N, algo=ra.PAL, policy=1, seed=9, max_episode_steps=40,
                    steps_per_launch=spl, **kw) as c:
        c.reset()
        st = c.train(K)
        same = np.all(np.abs(c.states.T - run.state) <= 1e-5 * (1 + np.abs(run.state)), axis=1) & (c.actions == run.action)
        assert same.mean() >= 0.95, same.mean()
        for i in np.flatnonzero(same)[:16]:
            assert np.max(np.abs(c.get_weights(i) - run.weights[i])) <= 5e-6
        assert abs(st["episodes"] - ost["episodes"]) <= max(2, 0.05 * ost["episodes"])
        assert abs(st["sum_abs_td_error"] - ost["sum_abs_td_error"]) <= 2e-3 * ost["sum_abs_td_error"]


def test_pal_fused_equals_stepwise(ra):
    kw = dict(n_envs=512, algo=5, policy=1, epsilon=0.1, lr=0.003, alpha=0.5, seed=3, max_episode_steps=60)
    with ra.Context(**kw) as c1, ra.Context(steps_per_launch=1, **kw) as c2:
        c1.reset(); c2.reset()
        c1.train(96); c2.train(96)
        assert np.array_equal(c1.states, c2.states) and np.array_equal(c1.actions, c2.actions)
        assert np.array_equal(c1.get_weights(7), c2.get_weights(7))


def test_pal_shared_weights(ra, orc):
    N, K = 600, 30
    kw = dict(gamma=0.9, lr=0.001 / 50, alpha=0.5, epsilon=0.1)
    ag = orc.make_agent(algo=orc.PAL, policy=orc.EGREEDY, shared_w=True, seed=17, max_episode_steps=25, **kw)
    run = orc.Run(ag, N, "f32")
    run.reset()
    run.train(K)
    with ra.Context(algo=ra.PAL, policy=1, weight_mode=ra.W_SHARED, seed=17, max_episode_steps=25, n_envs=N, **kw) as c:
        c.reset()
        c.train(K)
        Wd, Wo = c.get_weights(), run.weights
        assert np.max(np.abs(Wo)) > 1e-6
        assert np.max(np.abs(Wd - Wo)) <= 2e-5 * max(1.0, np.max(np.abs(Wo))) + 1e-7
