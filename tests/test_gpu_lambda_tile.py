"""GPU parity for the eligibility-trace agents on TILE CODING with per-learner tables (SURVEY 8f rank 1 widened off the register
family: the reference's traces are generic over the gradient buffer, rsrl/src/traces.rs:6-240; SARSALambda / QLambda
rsrl/src/control/td/sarsa_lambda.rs:53-98, q_lambda.rs:56-99).  One block per learner sweeps the learner's dense trace table
(rsrl_amd/csrc/kernels_lambda_tile.hpp); every element goes through the oracle's operations, so weights AND traces are
bit-identical to the CPU run in the device's arithmetic (f32d: the device's sincos polynomials restated)."""
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


@pytest.mark.parametrize("algo,trace,domain,T,B", [(3, 0, 1, 8, 8), (3, 1, 0, 4, 8), (4, 0, 1, 8, 8), (4, 2, 2, 4, 6), (3, 2, 1, 16, 4)])
def test_train_lambda_tile_bitwise(ra, orc, algo, trace, domain, T, B):
    # free-running driver loop, launches of 23 + 1 + 40 batch-steps (the launch boundary is invisible), step cap and terminals on the way
    N = 24
    kw = dict(gamma=0.99, alpha=0.05, lam=0.8, epsilon=0.2)
    ag = orc.make_agent(domain=domain, basis=orc.TILE, n_tilings=T, tiles_per_dim=B, algo=algo, policy=orc.EGREEDY, seed=11, trace=trace,
                        max_episode_steps=17, env_offset=5, **kw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    ost = run.train(64)
    with ra.Context(domain=domain, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=B, n_envs=N, algo=algo, policy=ra.EPSILON_GREEDY, seed=11,
                    trace=trace, max_episode_steps=17, env_offset=5, **kw) as c:
        c.reset()
        st = [c.train(k) for k in (23, 1, 40)]
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        for i in range(N):
            assert np.array_equal(c.get_weights(i), run.weights[i]), i
            assert np.array_equal(c.get_traces(i), run.traces[i]), i
        assert np.abs(run.weights).max() > 0 and np.abs(run.traces).max() > 0
        assert sum(s["episodes"] for s in st) == ost["episodes"] > 0
        assert sum(s["episodes_truncated"] for s in st) == ost["episodes_truncated"]
        assert sum(s["env_steps"] for s in st) == N * 64
        assert abs(sum(s["sum_abs_td_error"] for s in st) - ost["sum_abs_td_error"]) <= 1e-9 * ost["sum_abs_td_error"]


@pytest.mark.parametrize("algo,trace", [(3, 0), (3, 1), (4, 2)])
def test_handle_lambda_tile_bitwise(ra, orc, algo, trace):
    # Handler::handle on caller-supplied transitions from given (W, Z): TD error, weights and traces, bit for bit
    M, T, B = 40, 8, 8
    rng = np.random.default_rng(algo * 7 + trace)
    kw = dict(gamma=0.97, alpha=0.1, lam=0.9, epsilon=0.3)
    ag = orc.make_agent(domain=1, basis=orc.TILE, n_tilings=T, tiles_per_dim=B, algo=algo, policy=orc.EGREEDY, seed=4, trace=trace, **kw)
    s = rand_states(orc, 1, M, 21) * 0.5
    a = rng.integers(0, 2, M).astype(np.int32)
    with ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=B, n_envs=M, algo=algo, policy=1, seed=4, trace=trace, **kw) as c:
        F = c.F
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        term[::7] = 1
        Ws = (rng.normal(size=(M, F, 2)) * 0.1).astype(np.float32)
        Zs = (rng.normal(size=(M, F, 2)) * 0.4).astype(np.float32)
        Zs[rng.random((M, F, 2)) < 0.9] = 0.0                      # mostly-empty traces with a few live entries, some beyond +-1
        for i in range(M):
            c.set_weights(Ws[i], i)
            c.set_traces(Zs[i], i)
        assert np.array_equal(c.get_traces(3), Zs[3])
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            W, Z = Ws[i].copy(), Zs[i].copy()
            d = orc.handle_lambda(ag, W, Z, frm[:, i], a[i], rew[i], nxt[:, i], term[i], orc.draw(4, i, 0, orc.BLK_INNER), "f32d")
            assert td[i] == np.float32(d), (i, td[i], d)
            assert np.array_equal(c.get_traces(i), Z) and np.array_equal(c.get_weights(i), W), i


def test_lambda_tile_checkpoint_and_errors(ra, tmp_path):
    kw = dict(domain=1, basis=ra.TILE_CODING, n_tilings=4, tiles_per_dim=4, n_envs=16, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.2, alpha=0.05,
              lam=0.7, gamma=0.99, seed=3)
    path = str(tmp_path / "lt.bin")
    with ra.Context(**kw) as a:
        a.reset(); a.train(30, want_stats=False)
        a.save_weights(path)
        s, act = a.states, a.actions
        a.train(20, want_stats=False)
        ref = (a.get_weights(5), a.get_traces(5), a.states)
    with ra.Context(**kw) as b:
        b.load_weights(path)                                           # the trace tables travel with the checkpoint (aux kind 1)
        b.states, b.actions = s, act
        b.train(20, want_stats=False)
        assert np.array_equal(b.get_weights(5), ref[0]) and np.array_equal(b.get_traces(5), ref[1]) and np.array_equal(b.states, ref[2])
    with ra.Context(**dict(kw, weight_mode=ra.W_SHARED)) as sh:        # a shared table takes SPARSE per-learner traces instead (round 5,
        with pytest.raises(ra.RsrlHipError):                           # tests/test_gpu_sparse_lambda.py); those cannot be set from a dense matrix
            sh.set_traces(np.zeros((sh.F, sh.A), np.float32), 0)
