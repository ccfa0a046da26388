"""GPU parity for the eligibility-trace agents on the WAVE family (Fourier order 7 on the 4-D domains, F = 4096, one wavefront per
learner; rsrl_amd/csrc/kernels_wave_lambda.hpp).  SARSALambda / QLambda rsrl/src/control/td/sarsa_lambda.rs:53-98,
q_lambda.rs:56-99, trace rules rsrl/src/traces.rs:188-240.  The oracle's wave-order loop (orc_run_train_wave) restates the lane
partials and the DPP ladder of the dot products, so states, actions, weights and traces are compared bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


@pytest.mark.parametrize("algo,trace,domain,bf16", [(3, 0, 2, False), (3, 1, 1, False), (4, 2, 2, False), (4, 0, 1, False),
                                                     (3, 0, 2, True), (4, 1, 1, True), (3, 2, 1, True)])
def test_train_wave_lambda_bitwise(ra, orc, algo, trace, domain, bf16):
    # bf16 (round 6): W stored as bf16, every entry rounded stochastically at every step (Philox block 16 + 64 * column + lane), the trace f32
    N, K = 10, 40
    kw = dict(gamma=0.99, alpha=0.0005, lam=0.8, epsilon=0.2)
    ag = orc.make_agent(domain=domain, order=7, algo=algo, policy=orc.EGREEDY, seed=11, trace=trace, max_episode_steps=13, env_offset=5, **kw)
    run = orc.Run(ag, N, "f32d")
    run.reset_wave()
    ost = run.train_wave(K, bf16=bf16)
    with ra.Context(domain=domain, order=7, n_envs=N, algo=algo, policy=ra.EPSILON_GREEDY, seed=11, trace=trace, max_episode_steps=13,
                    env_offset=5, weight_dtype=ra.W_BF16 if bf16 else ra.W_F32, **kw) as c:
        c.reset()
        st = [c.train(k) for k in (17, 1, 22)]
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        for i in range(N):
            assert np.array_equal(c.get_weights(i), run.weights[i]), i
            assert np.array_equal(c.get_traces(i), run.traces[i]), i
        assert np.abs(run.weights).max() > 0 and np.abs(run.traces).max() > 0
        if bf16:
            assert np.all((run.weights.view(np.uint32) & 0xffff) == 0) and np.any((run.traces.view(np.uint32) & 0xffff) != 0)
        assert sum(s["episodes"] for s in st) == ost["episodes"] > 0
        assert sum(s["episodes_truncated"] for s in st) == ost["episodes_truncated"]
        assert sum(s["env_steps"] for s in st) == N * K
        assert abs(sum(s["sum_abs_td_error"] for s in st) - ost["sum_abs_td_error"]) <= 1e-6 * ost["sum_abs_td_error"]


@pytest.mark.parametrize("bf16", [False, True])
def test_handle_wave_lambda_equals_the_driver_loop_step(ra, bf16):
    # Handler::handle on the transitions the driver loop would have taken, from the same (W, Z): same TD bookkeeping, same bits
    N = 6
    kw = dict(domain=2, order=7, n_envs=N, algo=ra.SARSA_LAMBDA, policy=ra.EPSILON_GREEDY, epsilon=0.3, seed=2, gamma=0.98, alpha=0.001, lam=0.9,
              trace=ra.TRACE_SATURATE, max_episode_steps=1000, weight_dtype=ra.W_BF16 if bf16 else ra.W_F32)
    with ra.Context(**kw) as a, ra.Context(**kw) as b:
        a.reset(); b.reset()
        a.train(9); b.train(9)
        acts = b.actions.copy()
        frm, nxt, rew, term = b.domain_step(acts)                  # b's environments advance; its agent is taught by hand
        td = b.handle(frm, acts, rew, nxt, term)
        sa = a.train(1)
        assert np.array_equal(a.states, b.states)
        for i in range(N):
            assert np.array_equal(a.get_weights(i), b.get_weights(i)) and np.array_equal(a.get_traces(i), b.get_traces(i)), i
        assert abs(np.abs(td).sum() - sa["sum_abs_td_error"]) <= 1e-5 * sa["sum_abs_td_error"]


def test_wave_lambda_checkpoint_and_errors(ra, tmp_path):
    kw = dict(domain=1, order=7, n_envs=5, algo=ra.Q_LAMBDA, policy=ra.EPSILON_GREEDY, epsilon=0.2, alpha=0.0005, lam=0.7, gamma=0.99, seed=3,
              max_episode_steps=30)
    path = str(tmp_path / "wl.bin")
    with ra.Context(**kw) as c:
        c.reset(); c.train(25)
        c.save_weights(path)
        c.train(10)
        w_ref = [c.get_weights(i) for i in range(5)]; z_ref = [c.get_traces(i) for i in range(5)]
        states = c.states.copy()
    with ra.Context(**kw) as c:
        c.reset(); c.train(25)
        c.set_traces(np.zeros((4096, 2), np.float32), 2)
        c.load_weights(path)
        c.train(10)
        assert np.array_equal(c.states, states)
        for i in range(5):
            assert np.array_equal(c.get_weights(i), w_ref[i]) and np.array_equal(c.get_traces(i), z_ref[i])
    with ra.Context(**{**kw, "weight_dtype": ra.W_BF16}) as c, ra.Context(**{**kw, "weight_dtype": ra.W_BF16}) as d:      # bf16 weights travel as their f32 values
        c.reset(); c.train(25)
        p16 = str(tmp_path / "wl16.bin")
        c.save_weights(p16)
        d.reset(); d.load_weights(p16)
        d.states, d.actions = c.states, c.actions
        d.episode_steps = c.episode_steps
        c.train(10); d.train(10)
        assert np.array_equal(c.states, d.states)
        for i in range(5):
            assert np.array_equal(c.get_weights(i), d.get_weights(i)) and np.array_equal(c.get_traces(i), d.get_traces(i))
    with pytest.raises(ra.RsrlHipError):
        ra.Context(**{**kw, "weight_mode": ra.W_SHARED})
