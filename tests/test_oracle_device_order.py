"""CPU tests of the oracle's device-order leg: the "f32d" instantiation (device sincos / exp polynomials restated) and
orc_run_train_dev (carried phi / Q, rank-1 post-update Q) against the plain f32 / f64 oracles.  This is the CPU half of
the chain reference KATs -> f64 oracle -> f32 oracle -> f32d oracle == HIP path bitwise (tests/test_gpu_bitwise.py)."""
import numpy as np
import pytest


def _ulp_dist(a, b):
    a = np.asarray(a, dtype=np.float32); b = np.asarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def test_device_polynomials_close_to_libm(orc):
    # sincospi01 through the Fourier tables (order 1 => phi = [cos(pi s~_1), cos(pi s~_0), cos(pi(s~_0 + s~_1)), 1])
    rng = np.random.default_rng(0)
    lo, hi = orc.domain_bounds(0)
    worst = 0.0
    for _ in range(4000):
        s = lo + (hi - lo) * rng.random(2)
        worst = max(worst, np.abs(orc.fourier_project(0, 5, s, "f32d") - orc.fourier_project(0, 5, s, "f64")).max())
    assert worst <= 3e-6, worst                                  # the tolerance the device is held to (test_gpu_parity_mc.py)
    # sincos_cw through the dynamics: MountainCar cos(3x); CartPole / Acrobot RK4 (sin, cos of the angles, |x| <= ~30)
    for domain, n in ((0, 2000), (1, 500), (2, 500)):
        lo, hi = orc.domain_bounds(domain)
        for k in range(n):
            s = (lo + (hi - lo) * rng.random(len(lo))) * (0.5 if domain == 1 else 1.0)
            s = s.astype(np.float32)
            a = int(rng.integers(0, orc.lib().orc_domain_actions(domain)))
            nd, rd, td = orc.domain_step(domain, s, a, "f32d")
            n6, r6, t6 = orc.domain_step(domain, s.astype(np.float64), a, "f64")
            assert np.max(np.abs(nd - n6)) <= (2e-5 if domain == 2 else 1e-6) * (1 + np.max(np.abs(n6)))
    # exp_dev through the softmax probabilities (softmax.rs:15-37)
    for _ in range(3000):
        q = rng.normal(size=3) * rng.choice([0.1, 1.0, 10.0, 60.0])
        tau = float(rng.choice([0.3, 1.0, 2.5]))
        pd = orc.policy_probs(orc.SOFTMAX, q.astype(np.float32), tau=tau, prec="f32d")
        p6 = orc.policy_probs(orc.SOFTMAX, q.astype(np.float32).astype(np.float64), tau=np.float32(tau), prec="f64")
        assert np.max(np.abs(pd - p6)) <= 3e-7
        assert abs(pd.sum() - 1.0) <= 3e-7


@pytest.mark.parametrize("algo,policy", [(0, 1), (1, 1), (2, 1), (2, 2), (5, 1), (0, 0), (1, 2)])
def test_train_dev_follows_the_reference_order_loop(orc, algo, policy):
    # the device evaluation order (carried Q, rank-1 post-update Q, terminal -> s0 in the s' slot) walks the same
    # trajectories as the reference's order of operations (orc_run_train: 4 projections per step, Q re-evaluated from W)
    N, K = 48, 400
    kw = dict(gamma=0.9, lr=0.001, alpha=0.7, epsilon=0.1, tau=0.8)
    ag = orc.make_agent(algo=algo, policy=policy, seed=21, max_episode_steps=50, **kw)
    ref = orc.Run(ag, N, "f32"); ref.reset(); st_ref = ref.train(K)
    dev = orc.Run(ag, N, "f32"); dev.reset(); st_a = dev.train_dev(K // 3); st_b = dev.train_dev(K - K // 3)
    same = np.all(ref.state == dev.state, axis=1) & (ref.action == dev.action)
    assert same.mean() >= 0.95, same.mean()
    assert np.max(np.abs(ref.weights[same] - dev.weights[same])) <= 1e-6
    assert abs(st_a["episodes"] + st_b["episodes"] - st_ref["episodes"]) <= 2
    # ... and with the device's polynomials instead of libm (f32d), against the reference precision (f64)
    d2 = orc.Run(ag, N, "f32d"); d2.reset(); d2.train_dev(K)
    r64 = orc.Run(ag, N, "f64"); r64.reset(); r64.train(K)
    same = np.all(np.abs(r64.state - d2.state) <= 1e-4, axis=1) & (r64.action == d2.action)
    assert same.mean() >= 0.9, same.mean()
    assert np.max(np.abs(r64.weights[same] - d2.weights[same])) <= 2e-5


def test_train_dev_one_call_equals_many(orc):
    ag = orc.make_agent(policy=orc.EGREEDY, seed=3, max_episode_steps=40)
    a = orc.Run(ag, 32, "f32d"); a.reset(); a.train_dev(300)
    b = orc.Run(ag, 32, "f32d"); b.reset()
    for k in (1, 99, 200):
        b.train_dev(k)
    assert np.array_equal(a.state, b.state) and np.array_equal(a.action, b.action) and np.array_equal(a.weights, b.weights)
    with pytest.raises(ValueError):
        orc.Run(orc.make_agent(shared_w=True), 4, "f32d").train_dev(1)


def test_agent_policy_greedy_target_is_qlearning(orc):
    # ExpectedSARSA owning a Greedy policy (expected_sarsa.rs:22-29) == QLearning's delta (q_learning.rs:57-62) when the
    # maximum is unique; SARSA owning a Greedy policy bootstraps from the argmax
    rng = np.random.default_rng(1)
    lo, hi = orc.domain_bounds(0)
    kw = dict(gamma=0.95, lr=0.05, alpha=1.0, epsilon=0.3)
    es = orc.make_agent(algo=orc.EXPECTED_SARSA, policy=orc.EGREEDY, agent_policy=orc.GREEDY, **kw)
    sa = orc.make_agent(algo=orc.SARSA, policy=orc.EGREEDY, agent_policy=orc.GREEDY, **kw)
    ql = orc.make_agent(algo=orc.QLEARNING, policy=orc.EGREEDY, **kw)
    on = orc.make_agent(algo=orc.EXPECTED_SARSA, policy=orc.EGREEDY, **kw)
    assert on.apolicy == orc.EGREEDY and on.aeps_thr == on.eps_thr
    diff = 0.0
    for k in range(50):
        W = rng.normal(size=(36, 3)) * 0.3
        s, ns = lo + (hi - lo) * rng.random(2), lo + (hi - lo) * rng.random(2)
        a, term = int(rng.integers(0, 3)), int(k % 9 == 0)
        x = orc.draw(1, k, 0, orc.BLK_INNER)
        d = [orc.handle(g, W.copy(), s, a, -1.0, ns, term, x, "f64") for g in (es, sa, ql, on)]
        assert abs(d[0] - d[2]) <= 1e-12 and abs(d[1] - d[2]) <= 1e-12
        diff = max(diff, abs(d[3] - d[2]))
    assert diff > 1e-3


@pytest.mark.parametrize("domain,algo,policy", [(2, 2, 2), (1, 0, 1)])
def test_wave_order_loop_follows_the_reference_order_loop(orc, domain, algo, policy):
    # the wave family's evaluation order (feature index k = (f+1) mod F, lane partials + DPP ladder) against orc_run_train
    kw = dict(gamma=0.99, lr=0.001, alpha=1.0, epsilon=0.1, tau=1.0)
    ag = orc.make_agent(domain=domain, order=7, algo=algo, policy=policy, seed=23, max_episode_steps=15, **kw)
    N, K = 4, 24
    ref = orc.Run(ag, N, "f32"); ref.reset(); ref.train(K)
    wav = orc.Run(ag, N, "f32d"); wav.reset_wave(); wav.train_wave(K)
    assert np.all(np.abs(ref.state - wav.state) <= 1e-3 * (1 + np.abs(ref.state))) and np.array_equal(ref.action, wav.action)
    assert np.max(np.abs(ref.weights - wav.weights)) <= 1e-6
    b16 = orc.Run(ag, N, "f32d"); b16.reset_wave(); b16.train_wave(K, bf16=True)
    assert np.all((b16.weights.view(np.uint32) & 0xffff) == 0) and np.abs(b16.weights).max() > 0
    with pytest.raises(ValueError):
        orc.Run(orc.make_agent(order=5), 2, "f32d").train_wave(1)
