"""GPU parity for the prediction agents on TILE CODING (TD rsrl/src/prediction/td/td.rs:31-59, TDLambda
td_lambda.rs:41-78 on a ScalarLFA, fa/linear.rs:201-251, over TileCoding).  One block per learner
(rsrl_amd/csrc/kernels_td_tile.hpp); V(s), TD errors, weights and traces are bit-identical to the oracle in the device's
arithmetic."""
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


@pytest.mark.parametrize("algo,trace,domain,T,B", [(7, 0, 1, 8, 8), (7, 0, 0, 4, 8), (8, 0, 1, 8, 8), (8, 1, 0, 16, 8), (8, 2, 2, 4, 6)])
def test_train_td_tile_bitwise(ra, orc, algo, trace, domain, T, B):
    N = 24
    kw = dict(gamma=0.9, lr=0.05, alpha=0.05, lam=0.3)
    ag = orc.make_agent(domain=domain, basis=orc.TILE, n_tilings=T, tiles_per_dim=B, algo=algo, policy=orc.RANDOM, seed=11, trace=trace,
                        max_episode_steps=17, env_offset=5, **kw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    ost = run.train(64)
    with ra.Context(domain=domain, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=B, n_envs=N, algo=algo, policy=ra.RANDOM, seed=11,
                    trace=trace, max_episode_steps=17, env_offset=5, **kw) as c:
        assert c.n_out == 1
        c.reset()
        st = [c.train(k) for k in (23, 1, 40)]
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        for i in range(N):
            assert np.array_equal(c.get_weights(i).reshape(-1), run.weights[i].reshape(-1)), i
            if algo == 8:
                assert np.array_equal(c.get_traces(i).reshape(-1), run.traces[i].reshape(-1)), i
        assert np.abs(run.weights).max() > 0
        assert sum(s["episodes"] for s in st) == ost["episodes"] > 0
        assert sum(s["episodes_truncated"] for s in st) == ost["episodes_truncated"]
        assert sum(s["env_steps"] for s in st) == N * 64
        assert abs(sum(s["sum_abs_td_error"] for s in st) - ost["sum_abs_td_error"]) <= 1e-9 * ost["sum_abs_td_error"]


@pytest.mark.parametrize("algo,trace", [(7, 0), (8, 0), (8, 1), (8, 2)])
def test_handle_and_evaluate_td_tile_bitwise(ra, orc, algo, trace):
    M, T, B = 40, 8, 8
    rng = np.random.default_rng(algo * 7 + trace)
    kw = dict(gamma=0.97, lr=0.02, alpha=0.1, lam=0.9)
    ag = orc.make_agent(domain=1, basis=orc.TILE, n_tilings=T, tiles_per_dim=B, algo=algo, policy=orc.RANDOM, seed=4, trace=trace, **kw)
    s = rand_states(orc, 1, M, 21) * 0.5
    a = rng.integers(0, 2, M).astype(np.int32)
    with ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=B, n_envs=M, algo=algo, policy=ra.RANDOM, seed=4, trace=trace, **kw) as c:
        F = c.F
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        term[::7] = 1
        Ws = (rng.normal(size=(M, F)) * 0.1).astype(np.float32)
        Zs = (rng.normal(size=(M, F)) * 0.4).astype(np.float32)
        Zs[rng.random((M, F)) < 0.9] = 0.0
        for i in range(M):
            c.set_weights(Ws[i].reshape(F, 1), i)
            if algo == 8:
                c.set_traces(Zs[i].reshape(F, 1), i)
        v = c.q_evaluate(s)
        assert v.shape == (1, M)
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            assert v[0, i] == np.float32(orc.v_evaluate(ag, Ws[i], s[:, i], "f32d")), i
            W, Z = Ws[i].copy(), Zs[i].copy()
            d = orc.handle_td(ag, W, Z if algo == 8 else None, frm[:, i], rew[i], nxt[:, i], term[i], "f32d")
            assert td[i] == np.float32(d), (i, td[i], d)
            assert np.array_equal(c.get_weights(i).reshape(-1), W), i
            if algo == 8:
                assert np.array_equal(c.get_traces(i).reshape(-1), Z), i
        for call in (c.policy_mode, c.policy_probs, c.q_find_max):
            with pytest.raises(ra.RsrlHipError):
                call(s)


def test_td_tile_checkpoint_and_errors(ra, tmp_path):
    kw = dict(domain=1, basis=ra.TILE_CODING, n_tilings=4, tiles_per_dim=4, n_envs=16, algo=ra.TD_LAMBDA, policy=ra.RANDOM, alpha=0.05, lam=0.3,
              gamma=0.9, seed=3, max_episode_steps=30)
    path = str(tmp_path / "tdl.bin")
    with ra.Context(**kw) as c:
        c.reset(); c.train(50)
        c.save_weights(path)
        c.train(20)
        w_ref = [c.get_weights(i) for i in range(16)]; z_ref = [c.get_traces(i) for i in range(16)]
        states, actions = c.states.copy(), c.actions.copy()
    with ra.Context(**kw) as c:
        c.reset(); c.train(50)                                  # same seed, same counter: the environments are where they were
        c.load_weights(path)
        c.train(20)
        assert np.array_equal(c.states, states) and np.array_equal(c.actions, actions)
        for i in range(16):
            assert np.array_equal(c.get_weights(i), w_ref[i]) and np.array_equal(c.get_traces(i), z_ref[i])
    with pytest.raises(ra.RsrlHipError):
        ra.Context(**{**kw, "weight_mode": ra.W_SHARED})
    with pytest.raises(ra.RsrlHipError):
        ra.Context(**{**kw, "policy": 1})
