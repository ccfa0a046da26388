"""GPU parity for the wave family (one wavefront per learner): Fourier order 7 on the 4-D domains, F = 4096
(BASELINE.json configs[4]: Acrobot + ExpectedSARSA + Fourier(7) + Softmax, bf16 weights)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


def rand_states(orc, domain, M, seed, shrink=1.0):
    lo, hi = orc.domain_bounds(domain)
    rng = np.random.default_rng(seed)
    mid, half = (lo + hi) / 2, (hi - lo) / 2 * shrink
    return (mid[:, None] + half[:, None] * (2 * rng.random((len(lo), M)) - 1)).astype(np.float32)


@pytest.mark.parametrize("domain", [1, 2])
def test_project_order7(ra, orc, domain):
    M = 24
    s = rand_states(orc, domain, M, 7 + domain)
    with ra.Context(domain=domain, order=7, n_envs=M) as c:
        assert c.F == 4096
        phi = c.project(s)
    assert phi.shape == (4096, M) and np.all(phi[-1] == 1.0)       # constant feature last, as with_bias() stacks it
    for m in range(M):
        p64 = orc.fourier_project(domain, 7, s[:, m], "f64")
        p32 = orc.fourier_project(domain, 7, s[:, m], "f32")
        assert np.max(np.abs(phi[:, m] - p64)) <= 6e-6             # order 7, 4 dims: pi*28*ulp input rounding alone ~3e-6
        assert np.max(np.abs(phi[:, m] - p32)) <= 2e-6             # same op order; sincospi polynomial <= 1.7 ulp


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_weights_roundtrip_and_q(ra, orc, dtype):
    M = 6
    rng = np.random.default_rng(1)
    ag = orc.make_agent(domain=2, order=7)
    s = rand_states(orc, 2, M, 3)
    wd = ra.W_F32 if dtype == "f32" else ra.W_BF16
    with ra.Context(domain=2, order=7, n_envs=M, weight_dtype=wd) as c:
        Ws = [(rng.normal(size=(4096, 3)) * 0.05).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(Ws[i], i)
        back = [c.get_weights(i) for i in range(M)]
        q = c.q_evaluate(s)
        idx, val = c.q_find_max(s)
    for i in range(M):
        if dtype == "f32":
            assert np.array_equal(back[i], Ws[i])
        else:                                                       # stored rounded to bf16 (round to nearest even)
            assert np.max(np.abs(back[i] - Ws[i])) <= 2.0 ** -8 * np.max(np.abs(Ws[i]))
            assert np.all((back[i].view(np.uint32) & 0xffff) == 0)
        q64 = orc.q_evaluate(ag, back[i].astype(np.float64), s[:, i], "f64")
        assert np.allclose(q[:, i], q64, rtol=0, atol=5e-5 * (1 + np.abs(q64).max()))
        assert idx[i] == int(np.argmax(q[:, i])) or q[idx[i], i] == q[:, i].max()
        assert val[i] == q[:, i].max()


@pytest.mark.parametrize("domain,algo,policy", [(2, 2, 2), (1, 0, 1), (2, 1, 1)])
def test_handle_order7(ra, orc, domain, algo, policy):
    M = 10
    rng = np.random.default_rng(domain * 7 + algo)
    kw = dict(gamma=0.99, lr=0.001, alpha=0.5, epsilon=0.2, tau=1.0)
    ag = orc.make_agent(domain=domain, order=7, algo=algo, policy=policy, seed=3, **kw)
    s = rand_states(orc, domain, M, 11, shrink=0.5)
    A = 2 if domain == 1 else 3
    a = rng.integers(0, A, M).astype(np.int32)
    with ra.Context(domain=domain, order=7, algo=algo, policy=policy, seed=3, n_envs=M, **kw) as c:
        Ws = [(rng.normal(size=(4096, A)) * 0.02).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(Ws[i], i)
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            W = Ws[i].astype(np.float64)
            d = orc.handle(ag, W, frm[:, i], a[i], rew[i], nxt[:, i], term[i], orc.draw(3, i, 0, orc.BLK_INNER), "f64")
            # (SARSA's inner eps-greedy action is exact integer logic given equal Q: asserted like the other agents)
            assert abs(td[i] - d) <= 1e-4 * (1 + abs(d)), (i, td[i], d)
            assert np.max(np.abs(c.get_weights(i) - W)) <= 2e-6 * (1 + abs(d))


@pytest.mark.parametrize("domain,algo,policy", [(2, 2, 2), (1, 0, 1)])
def test_train_order7_vs_oracle_f32(ra, orc, domain, algo, policy):
    N, K = 12, 25
    kw = dict(gamma=0.99, lr=0.001, alpha=1.0, epsilon=0.1, tau=1.0)
    ag = orc.make_agent(domain=domain, order=7, algo=algo, policy=policy, seed=23, max_episode_steps=15, **kw)
    run = orc.Run(ag, N, "f32")
    run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, order=7, algo=algo, policy=policy, seed=23, max_episode_steps=15, n_envs=N, **kw) as c:
        c.reset()
        assert np.array_equal(c.actions, run.action.copy() * 0 + c.actions)     # shape sanity
        st = c.train(K)
        tol = 2e-3 if domain == 2 else 1e-5
        same = np.all(np.abs(c.states.T - run.state) <= tol * (1 + np.abs(run.state)), axis=1) & (c.actions == run.action)
        assert same.mean() >= 0.75, same.mean()
        for i in np.flatnonzero(same)[:4]:
            assert np.max(np.abs(c.get_weights(i) - run.weights[i])) <= (5e-4 if domain == 2 else 5e-6)
        assert st["env_steps"] == N * K and abs(st["episodes"] - ost["episodes"]) <= 2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_wave_fused_equals_stepwise_bitwise(ra, dtype):
    wd = ra.W_F32 if dtype == "f32" else ra.W_BF16
    kw = dict(domain=2, order=7, algo=2, policy=2, n_envs=9, seed=5, max_episode_steps=12, lr=0.01, gamma=0.99, weight_dtype=wd)
    with ra.Context(steps_per_launch=30, **kw) as a, ra.Context(steps_per_launch=1, **kw) as b, ra.Context(steps_per_launch=7, **kw) as d:
        for c in (a, b, d):
            c.reset()
            c.train(30)
        assert np.array_equal(a.states, b.states) and np.array_equal(a.states, d.states)
        assert np.array_equal(a.actions, b.actions)
        for i in (0, 8):
            assert np.array_equal(a.get_weights(i), b.get_weights(i)) and np.array_equal(a.get_weights(i), d.get_weights(i))
        if dtype == "bf16":
            assert np.all((a.get_weights(3).view(np.uint32) & 0xffff) == 0)
            assert np.max(np.abs(a.get_weights(3))) > 0


def test_bf16_stochastic_rounding_is_unbiased(ra):
    # updates far below bf16 resolution must survive on average (lr = 1e-3 problem, SURVEY 'hard parts'):
    # 400 identical tiny updates of one column; round-to-nearest would leave W untouched, SR moves it by the sum
    M = 4
    with ra.Context(domain=1, order=7, n_envs=M, weight_dtype=ra.W_BF16, lr=1.0, gamma=0.0, policy=0) as c:
        W0 = np.full((4096, 2), 1.0, dtype=np.float32)
        for i in range(M):
            c.set_weights(W0, i)
        s = np.zeros((4, M), dtype=np.float32)                      # phi(s0): every feature is cos(pi*c.s~) at s~ = 0.5
        a = np.zeros(M, dtype=np.int32)
        term = np.ones(M, dtype=np.uint8)                           # delta = r - Q(s,a)
        q0 = c.q_evaluate(s)[0]
        for k in range(400):
            q = c.q_evaluate(s)[0]
            r = (q + 1e-4).astype(np.float32)                       # delta = +1e-4 every time => column 0 += 1e-4 * phi
            c.handle(s, a, r, s, term)
        W = c.get_weights(0)
    phi = None
    with ra.Context(domain=1, order=7, n_envs=1) as c2:
        phi = c2.project(np.zeros((4, 1), dtype=np.float32))[:, 0]
    moved = W[:, 0] - 1.0
    big = np.abs(phi) > 0.5
    # expected drift 400 * 1e-4 * phi = 0.04 * phi; bf16 spacing near 1.0 is 2^-7 = 0.0078
    assert np.abs(np.mean(moved[big] / phi[big]) - 0.04) < 0.004
    assert np.array_equal(W[:, 1], W0[:, 1])
