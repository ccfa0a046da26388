"""GPU parity, bitwise, of the SURVEY 8f agents: SARSA(lambda) / Q(lambda) with the three trace rules, GreedyGQ, TD and
TDLambda against the oracle's "f32d" instantiation (device polynomials restated) driven by the reference-order loop
orc_run_train -- these kernels re-evaluate Q from the weights every step, exactly as the reference does, so no device-order
loop is needed: every learner's states, actions, weights and auxiliary matrix (trace / fa_td) must match bit for bit.
  sarsa_lambda.rs:53-98, q_lambda.rs:56-99, traces.rs:188-240, greedy_gq.rs:73-141, prediction/td/td.rs:31-59, td_lambda.rs:41-78"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


CASES = [
    ("sarsa(lambda) replacing", dict(domain=0, order=5, algo=3, policy=1, trace=1, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)),
    ("q(lambda) accumulating", dict(domain=0, order=5, algo=4, policy=1, trace=0, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)),
    ("sarsa(lambda) dutch, CartPole", dict(domain=1, order=1, algo=3, policy=1, trace=2, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)),
    ("sarsa(lambda) softmax, Acrobot", dict(domain=2, order=1, algo=3, policy=2, trace=0, gamma=0.99, alpha=0.005, lam=0.5, tau=1.0)),
    ("greedy_gq", dict(domain=0, order=3, algo=6, policy=1, gamma=0.99, lr=0.1, lr_td=0.001, epsilon=0.1)),
    ("td", dict(domain=0, order=5, algo=7, policy=3, gamma=0.99, lr=0.01)),
    ("td(lambda)", dict(domain=0, order=3, algo=8, policy=3, gamma=0.9, lam=0.3, trace=1)),
]


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_trace_gq_td_agents_bitwise(ra, orc, name, kw):
    N, K = 128, 400
    ag = orc.make_agent(seed=9, max_episode_steps=40, **kw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    ost = run.train(K)
    with ra.Context(n_envs=N, seed=9, max_episode_steps=40, **kw) as c:
        c.reset()
        st = c.train(150)
        st2 = c.train(K - 150)
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        for i in (0, 1, 63, 64, 127):
            assert np.array_equal(c.get_weights(i), run.weights[i]), i
            if kw["algo"] in (3, 4, 8):
                assert np.array_equal(c.get_traces(i), run.traces[i]), i
            if kw["algo"] == 6:
                assert np.array_equal(c.get_td_weights(i), run.traces[i]), i
        assert st["episodes"] + st2["episodes"] == ost["episodes"] > 0
    assert np.all(np.isfinite(run.weights)) and np.abs(run.weights).max() > 0
