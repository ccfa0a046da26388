"""GPU parity of QSigma (control/td/q_sigma.rs:80-202, with the documented one-line repair of Backup::propagate): the HIP
path through the C ABI against the oracle -- teacher-forced handle() sequences (f64 tolerance, f32d bitwise) and the
free-running driver loop (f32d: every learner bit for bit)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


@pytest.mark.parametrize("n_steps,sigma,policy", [(1, 0.0, 1), (3, 0.5, 1), (4, 1.0, 0), (2, 0.25, 2), (5, 0.5, 3)])
def test_qsigma_handle_sequences(ra, orc, n_steps, sigma, policy):
    M, K = 48, 30
    rng = np.random.default_rng(n_steps)
    lo, hi = orc.domain_bounds(0)
    kw = dict(gamma=0.9, lr=0.05, alpha=0.6, epsilon=0.2, tau=0.7)
    ag = orc.make_agent(algo=orc.Q_SIGMA, policy=policy, sigma=sigma, n_steps=n_steps, seed=4, **kw)
    W0 = [(rng.normal(size=(36, 3)) * 0.2).astype(np.float32) for _ in range(M)]
    with ra.Context(n_envs=M, algo=ra.Q_SIGMA, policy=policy, sigma=sigma, n_steps=n_steps, seed=4, **kw) as c:
        for i in range(M):
            c.set_weights(W0[i], i)
        Wd = [w.copy() for w in W0]                                   # f32d: bitwise
        W6 = [w.astype(np.float64) for w in W0]                      # f64: tolerance
        bd = [orc.QSigmaBackup(n_steps, "f32d") for _ in range(M)]
        b6 = [orc.QSigmaBackup(n_steps, "f64") for _ in range(M)]
        s = (lo[:, None] + (hi - lo)[:, None] * rng.random((2, M))).astype(np.float32)
        for k in range(K):
            a = rng.integers(0, 3, M).astype(np.int32)
            ns = (lo[:, None] + (hi - lo)[:, None] * rng.random((2, M))).astype(np.float32)
            term = (rng.random(M) < 0.1).astype(np.uint8)
            rew = -np.ones(M, dtype=np.float32)
            td = c.handle(s, a, rew, ns, term)
            for i in range(M):
                x = orc.draw(4, i, k, orc.BLK_INNER)
                d = bd[i].handle(ag, Wd[i], s[:, i], a[i], -1.0, ns[:, i], term[i], x)
                assert np.float32(d) == td[i], (k, i, d, td[i])
                if policy != 2:          # softmax: the inner a' may differ between exp implementations only at a cumulative-probability tie
                    d6 = b6[i].handle(ag, W6[i], s[:, i], a[i], -1.0, ns[:, i], term[i], x)
                    assert abs(td[i] - d6) <= 5e-5 * (1 + abs(d6))
            s = ns
        for i in range(M):
            w = c.get_weights(i)
            assert np.array_equal(w, Wd[i]), i
            if policy != 2:
                assert np.max(np.abs(w - W6[i])) <= 2e-5 * (1 + np.abs(W6[i]).max())
        assert max(np.abs(Wd[i] - W0[i]).max() for i in range(M)) > 1e-3


@pytest.mark.parametrize("domain,order,n_steps,sigma,policy", [(0, 5, 3, 0.5, 1), (0, 5, 1, 0.0, 1), (0, 3, 8, 0.0, 2), (0, 3, 8, 1.0, 2), (2, 1, 4, 0.5, 1), (1, 1, 2, 0.0, 0)])
def test_qsigma_free_running_bitwise(ra, orc, domain, order, n_steps, sigma, policy):
    N, K = 160, 700
    kw = dict(gamma=0.95, lr=0.01, alpha=0.8, epsilon=0.1, tau=1.0)
    ag = orc.make_agent(domain=domain, order=order, algo=orc.Q_SIGMA, policy=policy, sigma=sigma, n_steps=n_steps, seed=6, max_episode_steps=90, **kw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, order=order, n_envs=N, algo=ra.Q_SIGMA, policy=policy, sigma=sigma, n_steps=n_steps, seed=6,
                    max_episode_steps=90, **kw) as c:
        c.reset()
        st = c.train(300)
        st2 = c.train(K - 300)                      # the backup lives in device memory across launches
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        for i in (0, 1, 80, 159):
            # (Softmax with sigma > 0 divides by mu = the raw action value -- the reference's Function<(S, A)> of Softmax,
            # softmax.rs:84-92 -- and diverges; the device reproduces even that, NaN for NaN)
            assert np.array_equal(c.get_weights(i), run.weights[i], equal_nan=True), i
        if policy != 2 and domain == 0:      # (CartPole with sigma = 0: the terminal entry carries pi = 0, so z = 0 and its -1 never propagates;
            assert np.all(np.isfinite(run.weights)) and np.abs(run.weights).max() > 0      #  Softmax: 0 * (pi / mu) with mu = Q = 0 is NaN -- both as in the reference)
        assert st["episodes"] + st2["episodes"] == ost["episodes"] > 0
        n_d, _ = c.rollout_greedy(200)
    n_o, _ = run.rollout_greedy(200)
    assert np.array_equal(n_d, n_o)


def test_qsigma_rejects_what_it_cannot_run(ra):
    # (tile coding and the generic Fourier orders: built in round 4, tests/test_gpu_round4.py)
    for bad in (dict(weight_mode=ra.W_SHARED), dict(basis=ra.TILE_CODING, weight_mode=ra.W_SHARED), dict(sigma=1.5), dict(n_steps=0), dict(n_steps=33),
                dict(domain=0, order=3, weight_dtype=ra.W_BF16)):          # (bf16 weights: the order-7 wave family only, round 6 -- tests/test_gpu_wave_aux.py)
        with pytest.raises(ra.RsrlHipError):
            ra.Context(n_envs=8, algo=ra.Q_SIGMA, **bad)
