"""GPU parity for the prediction agents TD / TDLambda (rsrl/src/prediction/td/td.rs:31-59,
td_lambda.rs:41-78 on a ScalarLFA, rsrl/src/fa/linear.rs:201-251)."""
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


@pytest.mark.parametrize("domain,order,algo,trace", [(0, 5, 7, 0), (0, 5, 8, 0), (0, 3, 8, 1), (1, 1, 8, 2), (2, 1, 7, 0)])
def test_td_handle_and_evaluate(ra, orc, domain, order, algo, trace):
    M = 96
    rng = np.random.default_rng(domain * 3 + algo + trace)
    kw = dict(gamma=0.97, lr=0.02, alpha=0.1, lam=0.8)
    ag = orc.make_agent(domain=domain, order=order, algo=algo, policy=orc.RANDOM, seed=4, trace=trace, **kw)
    s = rand_states(orc, domain, M, 21)
    a = rng.integers(0, ag.n_actions, M).astype(np.int32)
    with ra.Context(domain=domain, order=order, n_envs=M, algo=algo, policy=ra.RANDOM, seed=4, trace=trace, **kw) as c:
        F = c.F
        assert c.n_out == 1 and c.A == ag.n_actions
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        term[::6] = 1
        ws = [(rng.normal(size=(F, 1)) * 0.2).astype(np.float32) for _ in range(M)]
        zs = [(rng.normal(size=(F, 1)) * 0.5).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(ws[i], i)
            if algo == 8:
                c.set_traces(zs[i], i)
        v = c.q_evaluate(s)                                    # Function<(S,)>::evaluate of the ScalarLFA
        assert v.shape == (1, M)
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            assert abs(v[0, i] - orc.v_evaluate(ag, ws[i].astype(np.float64), s[:, i])) <= 2e-6
            w, z = ws[i][:, 0].astype(np.float64), zs[i][:, 0].astype(np.float64)
            d = orc.handle_td(ag, w, z if algo == 8 else None, frm[:, i], rew[i], nxt[:, i], term[i])
            assert abs(td[i] - d) <= 2e-5 * (1 + abs(d)), (i, td[i], d)
            assert np.max(np.abs(c.get_weights(i)[:, 0] - w)) <= 3e-6 * (1 + abs(d))
            if algo == 8:
                assert np.max(np.abs(c.get_traces(i)[:, 0] - z)) <= 3e-6
        for call in (c.policy_mode, c.policy_probs, c.q_find_max):
            with pytest.raises(ra.RsrlHipError):
                call(s)


@pytest.mark.parametrize("domain,order,algo,kw", [
    (0, 5, 7, dict(gamma=0.99, lr=0.01)),
    (0, 3, 8, dict(gamma=0.9, lam=0.3, trace=1)),          # TDLambda steps with the raw TD error: short horizon keeps it bounded
    (1, 1, 7, dict(gamma=0.99, lr=0.05)),
])
def test_td_train_vs_oracle_f32(ra, orc, domain, order, algo, kw):
    N, K = 128, 60
    ag = orc.make_agent(domain=domain, order=order, algo=algo, policy=orc.RANDOM, seed=9, max_episode_steps=30, **kw)
    run = orc.Run(ag, N, "f32")
    run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, order=order, n_envs=N, algo=algo, policy=ra.RANDOM, seed=9, max_episode_steps=30, **kw) as c:
        c.reset()
        st = c.train(K)
        # a Random behaviour policy does not depend on the weights: states and actions must agree for EVERY learner
        assert np.array_equal(c.actions, run.action)
        assert np.max(np.abs(c.states.T - run.state)) <= 1e-5 * (1 + np.max(np.abs(run.state)))
        for i in range(0, N, 9):
            scale = max(1.0, np.abs(run.weights[i]).max())
            assert np.max(np.abs(c.get_weights(i) - run.weights[i])) <= 2e-4 * scale
        assert st["episodes"] == ost["episodes"] and st["env_steps"] == N * K
        assert abs(st["sum_abs_td_error"] - ost["sum_abs_td_error"]) <= 2e-3 * ost["sum_abs_td_error"]


@pytest.mark.parametrize("algo", [7, 8])
def test_td_fused_equals_stepwise(ra, algo):
    kw = dict(n_envs=400, order=3, algo=algo, policy=3, gamma=0.9, lr=0.01, lam=0.3, trace=1, seed=3, max_episode_steps=40)
    with ra.Context(**kw) as c1, ra.Context(steps_per_launch=1, **kw) as c2:
        c1.reset(); c2.reset()
        c1.train(90); c2.train(90)
        assert np.array_equal(c1.states, c2.states) and np.array_equal(c1.actions, c2.actions)
        assert np.array_equal(c1.get_weights(7), c2.get_weights(7))
        if algo == 8:
            assert np.array_equal(c1.get_traces(7), c2.get_traces(7))


def test_td_learns_the_value_of_the_random_policy(ra):
    # MountainCar under a random policy with a 100-step cap: every state is worth about -(1 - gamma^k)/(1 - gamma) < 0;
    # the mean prediction at the start state must move from 0 towards the discounted return of ~100 steps at -1
    with ra.Context(n_envs=2048, order=3, algo=7, policy=3, gamma=0.95, lr=0.005, seed=1, max_episode_steps=100) as c:
        c.reset()
        s0 = np.tile(np.array([[-0.5], [0.0]], dtype=np.float32), (1, 2048))
        assert np.all(c.q_evaluate(s0) == 0)
        c.train(3000)
        v = c.q_evaluate(s0)[0]
        target = -(1 - 0.95 ** 100) / (1 - 0.95)
        assert abs(v.mean() - target) < 0.25 * abs(target), (v.mean(), target)


def test_td_config_errors(ra):
    with pytest.raises(ra.RsrlHipError):
        ra.Context(n_envs=8, algo=7, policy=1)                       # no Q to be greedy on
    with pytest.raises(ra.RsrlHipError):
        ra.Context(n_envs=8, algo=7, policy=3, basis=ra.TILE_CODING, weight_mode=ra.W_SHARED)   # per-learner tables only
    with ra.Context(n_envs=8, algo=7, policy=3) as c:
        with pytest.raises(ra.RsrlHipError):
            c.get_traces(0)
        with pytest.raises(ra.RsrlHipError):
            c.rollout_greedy(10)
