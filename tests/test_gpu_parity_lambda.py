"""GPU parity for the eligibility-trace agents SARSA(lambda) / Q(lambda) (SURVEY 8f rank 1;
rsrl/src/control/td/sarsa_lambda.rs, q_lambda.rs, rsrl/src/traces.rs; driver rsrl/examples/sarsa_lambda.rs)."""
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


@pytest.mark.parametrize("algo,trace", [(3, 0), (3, 1), (3, 2), (4, 0), (4, 1)])
def test_handle_lambda(ra, orc, algo, trace):
    M = 96
    rng = np.random.default_rng(algo * 5 + trace)
    kw = dict(gamma=0.97, alpha=0.05, lam=0.8, epsilon=0.3)
    ag = orc.make_agent(algo=algo, policy=orc.EGREEDY, seed=4, trace=trace, **kw)
    s = rand_states(orc, 0, M, 21)
    a = rng.integers(0, 3, M).astype(np.int32)
    with ra.Context(n_envs=M, algo=algo, policy=1, seed=4, trace=trace, **kw) as c:
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        term[::7] = 1
        Ws = [(rng.normal(size=(36, 3)) * 0.1).astype(np.float32) for _ in range(M)]
        Zs = [(rng.normal(size=(36, 3)) * 0.4).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(Ws[i], i)
            c.set_traces(Zs[i], i)
        assert np.array_equal(c.get_traces(3), Zs[3])
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            W, Z = Ws[i].astype(np.float64), Zs[i].astype(np.float64)
            d = orc.handle_lambda(ag, W, Z, frm[:, i], a[i], rew[i], nxt[:, i], term[i], orc.draw(4, i, 0, orc.BLK_INNER))
            assert abs(td[i] - d) <= 3e-5 * (1 + abs(d)), (i, td[i], d)
            assert np.max(np.abs(c.get_traces(i) - Z)) <= 3e-6
            assert np.max(np.abs(c.get_weights(i) - W)) <= 3e-6 * (1 + abs(d))


@pytest.mark.parametrize("algo,trace,domain", [(3, 1, 0), (4, 0, 0), (3, 0, 1)])
def test_train_lambda_vs_oracle_f32(ra, orc, algo, trace, domain):
    # examples/sarsa_lambda.rs hyper-parameters: alpha 0.01, gamma 0.99, lambda 0.7, eps-greedy 0.2, replacing traces
    N, K = 128, 100
    order = 5 if domain == 0 else 1
    kw = dict(gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)
    ag = orc.make_agent(domain=domain, order=order, algo=algo, policy=orc.EGREEDY, seed=9, trace=trace, max_episode_steps=40, **kw)
    run = orc.Run(ag, N, "f32")
    run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, order=order, n_envs=N, algo=algo, policy=1, seed=9, trace=trace, max_episode_steps=40, **kw) as c:
        c.reset()
        st = c.train(K)
        same = np.all(np.abs(c.states.T - run.state) <= 1e-5, axis=1) & (c.actions == run.action)
        assert same.mean() >= 0.9, same.mean()
        for i in np.flatnonzero(same)[:12]:
            assert np.max(np.abs(c.get_weights(i) - run.weights[i])) <= 2e-5 * max(1.0, np.abs(run.weights[i]).max())
            assert np.max(np.abs(c.get_traces(i) - run.traces[i])) <= 1e-5
        assert abs(st["episodes"] - ost["episodes"]) <= max(2, 0.03 * ost["episodes"])


def test_lambda_fused_equals_stepwise_bitwise(ra):
    kw = dict(n_envs=700, algo=3, policy=1, epsilon=0.2, seed=1, gamma=0.99, alpha=0.01, lam=0.7, trace=1, max_episode_steps=30)
    with ra.Context(steps_per_launch=64, **kw) as a, ra.Context(steps_per_launch=1, **kw) as b:
        a.reset(), b.reset()
        sa, sb = a.train(64), b.train(64)
        assert np.array_equal(a.states, b.states) and np.array_equal(a.actions, b.actions)
        for i in (0, 350, 699):
            assert np.array_equal(a.get_weights(i), b.get_weights(i)) and np.array_equal(a.get_traces(i), b.get_traces(i))
        assert sa["episodes"] == sb["episodes"] > 0


def test_sarsa_lambda_learns_faster_than_q_learning(ra):
    # the reference's most developed MountainCar example (examples/sarsa_lambda.rs): replacing traces learn within
    # tens of thousands of steps where one-step Q-learning with SGD(0.001) still times out
    kw = dict(n_envs=2048, policy=1, epsilon=0.2, seed=0, max_episode_steps=1000)
    with ra.Context(algo=3, gamma=0.99, alpha=0.01, lam=0.7, trace=1, **kw) as c:
        c.reset()
        c.train(10000)
        st = c.train(10000)
    assert st["episodes"] > 0 and st["sum_episode_steps"] / st["episodes"] < 400


def test_lambda_error_paths(ra):
    with pytest.raises(ra.RsrlHipError):
        ra.Context(algo=3, weight_mode=ra.W_SHARED, n_envs=4)
    with ra.Context(algo=4, basis=ra.TILE_CODING, domain=1, n_envs=4):     # tile coding: built in round 3 (tests/test_gpu_lambda_tile.py)
        pass
    with ra.Context(algo=4, basis=ra.TILE_CODING, domain=1, n_envs=4, weight_mode=ra.W_SHARED):     # one shared table, sparse per-learner traces: round 5
        pass                                                                                        # (tests/test_gpu_sparse_lambda.py)
    with ra.Context(algo=3, domain=2, order=7, n_envs=4):                   # the order-7 wave family: built in round 3 (tests/test_gpu_wave_lambda.py)
        pass
    with ra.Context(algo=3, domain=2, order=7, n_envs=4, weight_dtype=ra.W_BF16):      # ... with bf16 tables + stochastic rounding since round 6
        pass
    with pytest.raises(ra.RsrlHipError):
        ra.Context(algo=3, lam=1.5, n_envs=4)
    with ra.Context(n_envs=4) as c:
        with pytest.raises(ra.RsrlHipError):
            c.get_traces(0)


# ---- SARSALambda / QLambda on the Fourier orders without a register-family kernel (round 5, rsrl_amd/csrc/kernels_lambda_mem.hpp): the reference's
# agents are generic over the approximator (sarsa_lambda.rs:37-52, q_lambda.rs:37-54).  Found missing by tests/fuzz_parity.py.
@pytest.mark.parametrize("domain,order,algo,trace,policy", [(0, 6, 3, 0, 1), (0, 7, 4, 2, 2), (1, 2, 3, 1, 1), (2, 3, 4, 0, 0), (1, 4, 3, 2, 2)])
def test_lambda_generic_fourier_orders_bitwise(ra, orc, domain, order, algo, trace, policy):
    N, K = 96, 120
    F = (order + 1) ** (2 if domain == 0 else 4)
    kw = dict(domain=domain, order=order, algo=algo, policy=policy, trace=trace, gamma=0.9, lam=0.9, alpha=0.3 / F, epsilon=0.2, tau=1.0, seed=21,
              max_episode_steps=40)
    ag = orc.make_agent(**kw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    ost = run.train(K)
    r64 = orc.Run(ag, N, "f64")
    r64.reset()
    r64.train(K)
    with ra.Context(n_envs=N, steps_per_launch=7, **kw) as c:
        assert c.F == F
        c.reset()
        st = [c.train(k) for k in (50, 1, K - 51)]                       # any split into launches (7 steps each)
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        for i in (0, 47, N - 1):
            assert np.array_equal(c.get_weights(i), run.weights[i]), i
            assert np.array_equal(c.get_traces(i), run.traces[i]), i
        assert sum(s["episodes"] for s in st) == ost["episodes"] > 0
        assert np.abs(run.weights).max() > 0 and np.abs(run.traces).max() > 0
        # the reference's precision: learners whose trajectory has not parted from the f64 run (an argmax decided by an fp32 rounding parts them)
        same = np.all(np.abs(c.states.T - r64.state) <= 1e-4 * (1 + np.abs(r64.state)), axis=1) & (c.actions == r64.action)
        assert same.mean() >= 0.8, same.mean()
        for i in np.flatnonzero(same)[:8]:
            assert np.max(np.abs(c.get_weights(i) - r64.weights[i])) <= 2e-5 * (1 + np.abs(r64.weights[i]).max())
        # Handler::handle on caller-supplied transitions: the same operations
        a0 = c.actions
        frm, nxt, rew, term = c.domain_step(a0)
        Wb = [c.get_weights(i).copy() for i in range(4)]
        Zb = [c.get_traces(i).copy() for i in range(4)]
        t_h = c.step_count                                                # the counter that keys the agent's own draw of this handle
        td = c.handle(frm, a0, rew, nxt, term)
        for i in range(4):
            W, Z = Wb[i].copy(), Zb[i].copy()
            d = orc.handle_lambda(ag, W, Z, frm[:, i], a0[i], rew[i], nxt[:, i], term[i], orc.draw(21, i, t_h, orc.BLK_INNER), "f32d")
            assert td[i] == np.float32(d), (i, td[i], d)
            assert np.array_equal(c.get_weights(i), W) and np.array_equal(c.get_traces(i), Z), i
