"""GPU tests of round 6's widenings inside SURVEY 8's rows (all through the C ABI):
  * the per-learner epsilon schedule (examples/sarsa_lambda.rs:48-75, :68) on the order-7 WAVE family -- the one-step agents with f32 and bf16 weights
    (k_train_wave / k_train_wave_pk <.., ESCHED>) and SARSALambda / QLambda (k_wave_lambda): bitwise against the oracle's wave-order loop;
  * rsrl_hip_measure_copy (bench.py's hbm_copy_measured)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


WAVE_SCHED = [
    ("sarsa f32, CartPole (agent shares the policy object)", dict(domain=1, order=7, algo=1, policy=1, gamma=0.99, lr=0.0005, epsilon=0.4), False, 0.95, 0.1),
    ("q-learning bf16, CartPole (packed registers)", dict(domain=1, order=7, algo=0, policy=1, gamma=0.99, lr=0.0005, epsilon=0.3), True, 0.9, 0.0),
    ("expected sarsa bf16, Acrobot", dict(domain=2, order=7, algo=2, policy=1, gamma=0.99, lr=0.00025, alpha=1.0, epsilon=0.3), True, 0.97, 0.0),
    ("sarsa(lambda), the reference example's schedule, CartPole", dict(domain=1, order=7, algo=3, policy=1, trace=1, gamma=0.99, alpha=0.0002, lam=0.7, epsilon=0.2), False, 0.995, 0.0),
    ("q(lambda), floor, Acrobot", dict(domain=2, order=7, algo=4, policy=1, trace=0, gamma=0.99, alpha=0.0001, lam=0.7, epsilon=0.3), False, 0.9, 0.05),
]


@pytest.mark.parametrize("name,kw,bf16,decay,floor", WAVE_SCHED, ids=[c[0] for c in WAVE_SCHED])
def test_epsilon_schedule_on_the_wave_family_bitwise(ra, orc, name, kw, bf16, decay, floor):
    N, K, cap = 24, 360, 12                                       # 30 episodes per learner at least
    ag = orc.make_agent(seed=9, max_episode_steps=cap, epsilon_decay=decay, epsilon_min=floor, **kw)
    run = orc.Run(ag, N, "f32d")
    run.reset_wave()
    ost = run.train_wave(K, bf16=bf16)
    with ra.Context(n_envs=N, seed=9, max_episode_steps=cap, epsilon_decay=decay, epsilon_min=floor, weight_dtype=ra.W_BF16 if bf16 else ra.W_F32, **kw) as c:
        c.reset()
        st = [c.train(k) for k in (1, 119, 240)]                  # any split into launches
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        assert np.array_equal(c.epsilons, run.eps.astype(np.float32))
        for i in (0, 1, 13, 23):
            assert np.array_equal(c.get_weights(i), run.weights[i]), i
            if kw["algo"] in (3, 4):
                assert np.array_equal(c.get_traces(i), run.traces[i]), i
        assert sum(s["episodes"] for s in st) == ost["episodes"] >= 30 * N
        # the schedule's state travels with a checkpoint and with policy_probs (every learner its own epsilon)
        e = c.epsilons
        p = c.policy_probs(c.states)
        assert np.allclose(p.min(axis=0), e / c.A, rtol=1e-6)
    assert run.eps.max() < np.float32(kw["epsilon"]) and run.eps.min() >= np.float32(floor)
    if floor > 0:
        assert (run.eps == np.float32(floor)).any()


def test_wave_family_schedule_is_refused_where_no_kernel_runs_it(ra):
    for bad in (dict(algo=6, lr_td=0.01), dict(algo=9, n_steps=2, sigma=0.5), dict(algo=7, policy=3), dict(algo=0, policy=2)):
        with pytest.raises(ra.RsrlHipError):
            ra.Context(**{**dict(n_envs=8, domain=2, order=7, policy=1, epsilon_decay=0.9), **bad})


def test_measure_copy(ra):
    out = C.c_double()
    L = ra._abi.lib()
    assert L.rsrl_hip_measure_copy(0, 1 << 28, 5, C.byref(out)) == 0
    assert 1000.0 < out.value < 8000.0, out.value                 # GB/s, read + write: between a quarter of and the published peak
    assert L.rsrl_hip_measure_copy(0, 8, 5, C.byref(out)) == -1 and L.rsrl_hip_measure_copy(0, 1 << 20, 0, C.byref(out)) == -1
    assert L.rsrl_hip_measure_copy(99, 1 << 20, 1, C.byref(out)) == -2
