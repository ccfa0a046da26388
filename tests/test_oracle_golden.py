"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md §8c).  Each test cites the reference test it transcribes (paths under /root/reference)."""
import math

import numpy as np
import pytest


# ---------------------------------------------------------------------------- RNG pin
def test_philox_known_answers(orc):
    # Random123 kat_vectors for philox4x32-10
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


# ---------------------------------------------------------------------------- domains
def test_cartpole_initial_observation(orc):
    # rsrl_domains/src/cart_pole.rs:128-141
    s = orc.domain_reset(orc.CART_POLE)
    assert list(s) == [0.0, 0.0, 0.0, 0.0]
    assert not orc.domain_is_terminal(orc.CART_POLE, s)


@pytest.mark.parametrize("action,sign", [(0, 1.0), (1, -1.0)])
def test_cartpole_step(orc, action, sign):
    # rsrl_domains/src/cart_pole.rs:143-162 (test_step_0) and :164-183 (test_step_1), tol 1e-7
    s = orc.domain_reset(orc.CART_POLE)
    s, r, term = orc.domain_step(orc.CART_POLE, s, action)
    exp1 = np.array([-0.0032931628891235, -0.3293940797883472, 0.0029499634056967, 0.2951522145037250])
    assert np.all(np.abs(s - sign * exp1) < 1e-7)
    assert np.all(np.abs(s - sign * exp1) < 1e-15)          # the restatement reproduces all printed digits
    assert r == 0.0 and not term
    s, r, term = orc.domain_step(orc.CART_POLE, s, action)
    exp2 = np.array([-0.0131819582085161, -0.6597158115002169, 0.0118185373734479, 0.5921703414056713])
    assert np.all(np.abs(s - sign * exp2) < 1e-7)
    assert np.all(np.abs(s - sign * exp2) < 1e-15)


def test_cartpole_step_f32_close(orc):
    s = orc.domain_reset(orc.CART_POLE, "f32")
    s, _, _ = orc.domain_step(orc.CART_POLE, s, 0, "f32")
    exp1 = np.array([-0.0032931628891235, -0.3293940797883472, 0.0029499634056967, 0.2951522145037250])
    assert np.all(np.abs(s - exp1) < 1e-7)


def test_acrobot_initial_observation(orc):
    # rsrl_domains/src/acrobot.rs:159-174
    s = orc.domain_reset(orc.ACROBOT)
    assert list(s) == [0.0, 0.0, 0.0, 0.0]
    assert not orc.domain_is_terminal(orc.ACROBOT, s)


def test_mountain_car_initial_observation(orc):
    # rsrl_domains/src/mountain_car/discrete.rs:109-120
    s = orc.domain_reset(orc.MOUNTAIN_CAR)
    assert s[0] == -0.5 and s[1] == 0.0
    assert not orc.domain_is_terminal(orc.MOUNTAIN_CAR, s)


def test_mountain_car_is_terminal(orc):
    # rsrl_domains/src/mountain_car/discrete.rs:122-137
    X_MAX = 0.6
    T = lambda x, v: orc.domain_is_terminal(orc.MOUNTAIN_CAR, [x, v])
    assert not T(-0.5, 0.0)
    assert T(X_MAX, -0.05) and T(X_MAX, 0.0) and T(X_MAX, 0.05)
    assert not T(X_MAX - 0.0001 * X_MAX, 0.0)
    assert T(X_MAX + 0.0001 * X_MAX, 0.0)


def test_mountain_car_restatement_vectors(orc):
    # SURVEY.md Appendix C.2 (restatement-derived, literal f64 Python restatement of discrete.rs:58-65)
    exp = [(-4.99176843004169257e-01, 8.23156995830742755e-04),
           (-4.97536686679353246e-01, 1.64015632481602459e-03),
           (-4.97091796932347396e-01, 4.44889747005862727e-04),
           (-4.96845500067847445e-01, 2.46296864499942676e-04)]
    s = orc.domain_reset(orc.MOUNTAIN_CAR)
    for a, (x, v) in zip([2, 2, 0, 1], exp):
        s, r, term = orc.domain_step(orc.MOUNTAIN_CAR, s, a)
        assert abs(s[0] - x) < 1e-12 and abs(s[1] - v) < 1e-12
        assert r == -1.0 and not term


def test_mountain_car_independent_python_restatement(orc):
    # literal Python f64 restatement of discrete.rs:58-65 compared over a random walk
    rng = np.random.default_rng(1)
    x, v = -0.5, 0.0
    s = orc.domain_reset(orc.MOUNTAIN_CAR)
    for _ in range(500):
        a = int(rng.integers(0, 3))
        dv = 0.001 * (a - 1) + -0.0025 * math.cos(3.0 * x)      # dv() is evaluated first (discrete.rs:58)
        v = max(-0.07, min(0.07, v + dv))
        x = max(-1.2, min(0.6, x + v))
        s, r, term = orc.domain_step(orc.MOUNTAIN_CAR, s, a)
        assert s[0] == x and s[1] == v
        assert term == (x >= 0.6)
        assert r == (0.0 if term else -1.0)
        if term:
            break


def test_acrobot_restatement_vectors(orc):
    # SURVEY.md Appendix C.2 (restatement-derived from acrobot.rs:60-108)
    exp = [[3.5921786557804793e-02, -1.4429803220337406e-02, 3.6724547420740000e-01, -1.5201763594674830e-01],
           [1.5389994361268919e-01, -6.7375050379204354e-02, 8.4156643576009782e-01, -4.0382076982334458e-01],
           [3.1745962403996486e-01, -1.6211248822662494e-01, 8.3742037169521832e-01, -5.8047932886129550e-01],
           [5.4327733779203724e-01, -3.2857127502458527e-01, 1.4878169406568105e+00, -1.1332218900869830e+00]]
    s = orc.domain_reset(orc.ACROBOT)
    for a, e in zip([2, 2, 0, 1], exp):
        s, r, term = orc.domain_step(orc.ACROBOT, s, a)
        assert np.all(np.abs(s - np.array(e)) < 1e-12)
        assert r == -1.0 and not term


def test_cartpole_terminal_sits_on_bound(orc):
    # cart_pole.rs:44-49 clip + :87-90 `<=`/`>=`: pushing right forever ends exactly on a limit, reward -1
    s = orc.domain_reset(orc.CART_POLE)
    for _ in range(500):
        s, r, term = orc.domain_step(orc.CART_POLE, s, 1)
        if term:
            break
    assert term and r == -1.0
    assert (abs(s[0]) == 2.4) or (abs(s[2]) == math.pi / 15.0)


# ---------------------------------------------------------------------------- utils / policies
def test_greedy_sample_cases(orc):
    # rsrl/src/policies/greedy.rs:96-145 (MockQ echoes the state as the Q-vector)
    S = lambda q: orc.policy_sample(orc.GREEDY, q, (0, 0, 0, 0))
    assert S([1.0]) == 0 and S([-100.0]) == 0                              # test_1d
    assert S([10.0, 1.0]) == 0 and S([1.0, 10.0]) == 1                     # test_two_positive
    assert S([-10.0, -1.0]) == 1 and S([-1.0, -10.0]) == 0                 # test_two_negative
    assert S([10.0, -1.0]) == 0 and S([-10.0, 1.0]) == 1                   # test_two_alt
    assert S([1.0, -10.0]) == 0 and S([-1.0, 10.0]) == 1
    assert S([-123.1, 123.1, 250.5, -1240.0, -4500.0, 10000.0, 20.1]) == 5  # test_long


def test_greedy_precision(orc):
    # rsrl/src/policies/greedy.rs:147-153: in f64 2e-7 - 1e-7 == 1e-7 exactly is NOT < 1e-7
    for x in [(0, 0, 0, 0), (0, 0, 0xffffffff, 0)]:
        assert orc.policy_sample(orc.GREEDY, [1e-7, 2e-7], x) == 1


def test_greedy_probabilities(orc):
    # rsrl/src/policies/greedy.rs:155-168, tol 1e-6
    p = orc.policy_probs(orc.GREEDY, [1e-7, 1e-7, 1e-7, 1e-7])
    assert np.allclose(p, [0.25] * 4, atol=1e-6)
    p = orc.policy_probs(orc.GREEDY, [1e-7, 2e-7, 3e-7, 4e-7])
    assert np.allclose(p, [0.0, 0.0, 0.0, 1.0], atol=1e-6)


def test_argmaxima_semantics(orc):
    # rsrl/src/utils.rs:6-21: near-ties do not raise the running max
    ix, mx = orc.argmaxima([1.0, 1.0 + 5e-8, 1.0 + 9e-8, 1.0 + 1.5e-7])
    assert ix == [3] and mx == 1.0 + 1.5e-7
    ix, mx = orc.argmaxima([1.0, 1.0 + 5e-8, 1.0 + 9e-8])
    assert ix == [0, 1, 2] and mx == 1.0
    assert orc.find_max([3.0, 1.0, 3.0])[0] == 2          # core.rs:96-105 ties -> last
    assert orc.find_max([3.0, 1.0, 2.0])[0] == 0
    assert orc.argmax_first([0.2, 0.5, 0.5]) == 1         # utils.rs:23-34 ties -> first


def test_egreedy_probabilities(orc):
    # rsrl/src/policies/epsilon_greedy.rs:115-133, eps=0.5, A=5, tol 1e-6
    P = lambda q: orc.policy_probs(orc.EGREEDY, q, eps=0.5)
    assert np.allclose(P([1.0, 0.0, 0.0, 0.0, 0.0]), [0.6, 0.1, 0.1, 0.1, 0.1], atol=1e-6)
    assert np.allclose(P([0.0, 0.0, 0.0, 0.0, 1.0]), [0.1, 0.1, 0.1, 0.1, 0.6], atol=1e-6)
    assert np.allclose(P([1.0, 0.0, 0.0, 0.0, 1.0]), [0.35, 0.1, 0.1, 0.1, 0.35], atol=1e-6)


def test_egreedy_probabilities_uniform(orc):
    # rsrl/src/policies/epsilon_greedy.rs:135-145
    assert np.allclose(orc.policy_probs(orc.EGREEDY, [-1.0, 0.0, 0.0, 0.0], eps=1.0), [0.25] * 4, atol=1e-6)


def test_egreedy_sampling_frequencies(orc):
    # rsrl/src/policies/epsilon_greedy.rs:95-113: Q=[1,0], eps=0.5 -> 0.75/0.25 +- 0.05 over 10^4 draws
    n0 = 0
    for t in range(10000):
        x = orc.draw(7, 0, t, orc.BLK_STEP)
        n0 += orc.policy_sample(orc.EGREEDY, [1.0, 0.0], x, eps=0.5) == 0
    assert abs(0.75 - n0 / 10000.0) < 0.05


def test_random_sampling_frequencies(orc):
    # rsrl/src/policies/random.rs:58-76: A=2 -> 0.5/0.5 +- 0.05
    n0 = 0
    for t in range(10000):
        x = orc.draw(11, 3, t, orc.BLK_STEP)
        n0 += orc.policy_sample(orc.RANDOM, [1.0, 0.0], x) == 0
    assert abs(0.5 - n0 / 10000.0) < 0.05
    assert orc.policy_mode(orc.RANDOM, [1.0, 0.0]) == -1        # random.rs:47 `mode` panics


def test_softmax_cases(orc):
    # rsrl/src/policies/softmax.rs:249-256 (test_1d): single action -> always 0
    for i in range(1, 100):
        assert orc.policy_sample(orc.SOFTMAX, [float(i)], orc.draw(0, 0, i, 0)) == 0
    # documented intent (disabled tests softmax.rs:273-291)
    E = math.e
    assert np.allclose(orc.policy_probs(orc.SOFTMAX, [0.0, 1.0]), [1 / (1 + E), E / (1 + E)], atol=1e-6)
    assert np.allclose(orc.policy_probs(orc.SOFTMAX, [0.0, 2.0]), [1 / (1 + E * E), E * E / (1 + E * E)],
                       atol=1e-6)
    # softmax.rs:258-271 intent: sampling frequencies for Q=[0,1]
    c1 = sum(orc.policy_sample(orc.SOFTMAX, [0.0, 1.0], orc.draw(3, 1, t, 0)) for t in range(20000))
    assert abs(c1 / 20000.0 - E / (1 + E)) < 1e-2
    # mode = argmax_first over the probabilities (softmax.rs:141-143)
    assert orc.policy_mode(orc.SOFTMAX, [0.0, 2.0, 2.0]) == 1


# ---------------------------------------------------------------------------- Fourier / TD (restatement-derived)
def test_fourier_layout(orc):
    # F=(order+1)^D, bias last, row k <-> c=(k div 6, k mod 6), all-zero skipped
    phi = orc.fourier_project(orc.MOUNTAIN_CAR, 5, [-0.5, 0.0])
    assert phi.shape == (36,) and phi[-1] == 1.0
    xs, vs = (-0.5 - -1.2) / (0.6 - -1.2), (0.0 - -0.07) / (0.07 - -0.07)
    for k in range(1, 36):
        c0, c1 = divmod(k, 6)
        assert abs(phi[k - 1] - math.cos(math.pi * (c0 * xs + c1 * vs))) < 1e-14
    phi7 = orc.fourier_project(orc.ACROBOT, 7, [0.1, -0.2, 0.3, -0.4])
    assert phi7.shape == (4096,) and phi7[-1] == 1.0


def test_qlearning_single_update_matches_hand_computation(orc):
    # q_learning.rs:51-71 + fa/linear.rs:379-391 restated by hand in numpy
    ag = orc.make_agent(policy=orc.GREEDY, gamma=0.9, lr=0.001)
    rng = np.random.default_rng(0)
    W = rng.normal(size=(36, 3)) * 0.1
    s = np.array([-0.6, 0.01])
    ns, r, term = orc.domain_step(orc.MOUNTAIN_CAR, s, 2)
    phi, nphi = orc.fourier_project(0, 5, s), orc.fourier_project(0, 5, ns)
    delta_exp = r + 0.9 * np.max(nphi @ W) - phi @ W[:, 2]
    W_exp = W.copy()
    W_exp[:, 2] += 0.001 * delta_exp * phi
    W2 = W.copy()
    delta = orc.handle(ag, W2, s, 2, r, ns, term)
    assert abs(delta - delta_exp) < 1e-12
    assert np.max(np.abs(W2 - W_exp)) < 1e-14
    # terminal transition: delta = r - Q(s,a)   (q_learning.rs:55-56)
    W3 = W.copy()
    d = orc.handle(ag, W3, s, 1, 0.0, ns, True)
    assert abs(d - (0.0 - phi @ W[:, 1])) < 1e-12


def test_expected_sarsa_and_sarsa_single_update(orc):
    # expected_sarsa.rs:45-66 (error = alpha*delta) and sarsa.rs:53-75 (inner policy sample)
    rng = np.random.default_rng(2)
    W = rng.normal(size=(36, 3)) * 0.1
    s = np.array([-0.4, -0.02])
    ns, r, term = orc.domain_step(orc.MOUNTAIN_CAR, s, 0)
    phi, nphi = orc.fourier_project(0, 5, s), orc.fourier_project(0, 5, ns)
    qn = nphi @ W
    ag = orc.make_agent(algo=orc.EXPECTED_SARSA, policy=orc.EGREEDY, epsilon=0.2, gamma=0.95, lr=0.01, alpha=0.5)
    p = orc.policy_probs(orc.EGREEDY, qn, eps=0.2)
    delta_exp = r + 0.95 * float(qn @ p) - phi @ W[:, 0]
    W2 = W.copy()
    d = orc.handle(ag, W2, s, 0, r, ns, term)
    assert abs(d - delta_exp) < 1e-12
    W_exp = W.copy()
    W_exp[:, 0] += 0.01 * 0.5 * delta_exp * phi
    assert np.max(np.abs(W2 - W_exp)) < 1e-14
    ag = orc.make_agent(algo=orc.SARSA, policy=orc.EGREEDY, epsilon=0.2, gamma=0.95, lr=0.01)
    x = orc.draw(5, 9, 3, orc.BLK_INNER)
    na = orc.policy_sample(orc.EGREEDY, qn, x, eps=0.2)
    W2 = W.copy()
    d = orc.handle(ag, W2, s, 0, r, ns, term, x_inner=x)
    assert abs(d - (r + 0.95 * qn[na] - phi @ W[:, 0])) < 1e-12


def test_driver_loop_n1_matches_manual_composition(orc):
    # examples/q_learning.rs:34-55 composed by hand from the single-call oracle functions
    ag = orc.make_agent(policy=orc.EGREEDY, epsilon=0.1, seed=3, max_episode_steps=50)
    run = orc.Run(ag, 1)
    run.reset()
    W = np.zeros((36, 3))
    s = orc.domain_reset(0)
    a = orc.policy_sample(orc.EGREEDY, orc.q_evaluate(ag, W, s), orc.draw(3, 0, 0, orc.BLK_INIT), eps=0.1)
    assert a == run.action[0]
    ep = 0
    for t in range(120):
        ns, r, term = orc.domain_step(0, s, a)
        orc.handle(ag, W, s, a, r, ns, term)
        a = orc.policy_sample(orc.EGREEDY, orc.q_evaluate(ag, W, ns), orc.draw(3, 0, t, orc.BLK_STEP), eps=0.1)
        ep += 1
        if term or ep >= 50:
            ns = orc.domain_reset(0)
            a = orc.policy_sample(orc.EGREEDY, orc.q_evaluate(ag, W, ns), orc.draw(3, 0, t, orc.BLK_RESET),
                                  eps=0.1)
            ep = 0
        s = ns
        run.train(1)
        assert np.array_equal(run.state[0], s) and run.action[0] == a and run.ep_step[0] == ep
    assert np.array_equal(run.weights[0], W)


def test_shared_w_n1_equals_per_env(orc):
    # the synchronous mini-batch rule collapses to the reference rule at N=1
    a1 = orc.make_agent(policy=orc.EGREEDY, seed=4, shared_w=False)
    a2 = orc.make_agent(policy=orc.EGREEDY, seed=4, shared_w=True)
    r1, r2 = orc.Run(a1, 1), orc.Run(a2, 1)
    r1.reset(), r2.reset()
    r1.train(300), r2.train(300)
    assert np.allclose(r1.weights[0], r2.weights, rtol=0, atol=1e-15)
    assert np.array_equal(r1.state, r2.state)


def test_rollout_semantics(orc):
    # rsrl_domains/src/lib.rs:448-479 + :340: n_states = 1 + min(limit-1, T)
    ag = orc.make_agent(policy=orc.GREEDY)
    run = orc.Run(ag, 2)
    n, tot = run.rollout_greedy(500)            # W = 0: mode -> last action (push right) forever, never reaches goal
    assert list(n) == [500, 500] and list(tot) == [-499.0, -499.0]
    n, _ = run.rollout_greedy(1)
    assert list(n) == [1, 1]
    with pytest.raises(ValueError):
        run.rollout_greedy(0)


def test_tile_indices_definition(orc):
    # SURVEY.md Appendix B.3 restated in numpy float32
    ag = orc.make_agent(domain=orc.CART_POLE, basis=orc.TILE, n_tilings=8, tiles_per_dim=8)
    lo, hi = orc.domain_bounds(orc.CART_POLE)
    lo, hi = lo.astype(np.float32), hi.astype(np.float32)
    rng = np.random.default_rng(5)
    for _ in range(200):
        s = (lo + (hi - lo) * rng.random(4).astype(np.float32)).astype(np.float32)
        idx = orc.tile_indices(ag, s)
        for t in range(8):
            lin = 0
            for i in range(4):
                sc = np.float32(np.float32(s[i] - lo[i]) / np.float32(hi[i] - lo[i]))
                u = np.float32(sc * np.float32(7))
                off = np.float32(np.float32((t * (2 * i + 1)) % 8) / np.float32(8))
                cell = int(np.floor(np.float32(u + off)))
                lin += min(7, max(0, cell)) * 8 ** i
            assert idx[t] == t * 4096 + lin
    assert orc.n_features(ag) == 32768


# ---------------------------------------------------------------------------- eligibility traces
def test_trace_rules_reference_doctest(orc):
    # rsrl/src/traces.rs:112-126 (doctest): Accumulate{gamma 0.95, lambda 0.7}: 1.0 -> 0.665 after an all-zero update.
    # Driven through the agent: the constant feature (value 1) of column a is the probe.
    ag = orc.make_agent(algo=orc.SARSA_LAMBDA, policy=orc.GREEDY, gamma=0.95, lam=0.7, alpha=0.0, trace=orc.TRACE_ACCUMULATE)
    W, Z = np.zeros((36, 3)), np.zeros((36, 3))
    s = np.array([-0.5, 0.0]); ns = np.array([-0.49, 0.001])
    orc.handle_lambda(ag, W, Z, s, 0, -1.0, ns, False)
    assert Z[35, 0] == 1.0 and Z[35, 1] == 0.0
    orc.handle_lambda(ag, W, Z, s, 1, -1.0, ns, False)
    assert abs(Z[35, 0] - 0.665) < 1e-12 and Z[35, 1] == 1.0        # column 0 received a zero gradient
    orc.handle_lambda(ag, W, Z, s, 1, -1.0, ns, True)                # terminal: trace.reset()  (sarsa_lambda.rs:76)
    assert np.all(Z == 0.0)
    # Saturate (Trace::replacing) clips at +-1, Dutch multiplies the decay by (1 - alpha)   (traces.rs:205-240)
    ag = orc.make_agent(algo=orc.SARSA_LAMBDA, policy=orc.GREEDY, gamma=1.0, lam=1.0, alpha=0.0, trace=orc.TRACE_SATURATE)
    Z[:] = 0
    for _ in range(3):
        orc.handle_lambda(ag, W, Z, s, 2, -1.0, ns, False)
    assert Z[35, 2] == 1.0 and np.all(np.abs(Z) <= 1.0)
    ag = orc.make_agent(algo=orc.SARSA_LAMBDA, policy=orc.GREEDY, gamma=0.9, lam=0.5, alpha=0.2, trace=orc.TRACE_DUTCH)
    Z[:] = 0
    orc.handle_lambda(ag, W, Z, s, 0, -1.0, ns, False)
    orc.handle_lambda(ag, W, Z, s, 1, -1.0, ns, False)
    assert abs(Z[35, 0] - 0.9 * 0.5 * 0.8) < 1e-12


def test_lambda_agents_hand_computation(orc):
    # sarsa_lambda.rs:53-98 / q_lambda.rs:56-99 restated in numpy for one transition
    rng = np.random.default_rng(5)
    W0 = rng.normal(size=(36, 3)) * 0.1
    Z0 = rng.normal(size=(36, 3)) * 0.05
    s = np.array([-0.7, 0.02])
    ns, r, term = orc.domain_step(0, s, 1)
    phi, nphi = orc.fourier_project(0, 5, s), orc.fourier_project(0, 5, ns)
    q, qn = phi @ W0, nphi @ W0
    for algo in (orc.SARSA_LAMBDA, orc.Q_LAMBDA):
        ag = orc.make_agent(algo=algo, policy=orc.EGREEDY, epsilon=0.3, gamma=0.97, alpha=0.05, lam=0.8, seed=2)
        x = orc.draw(2, 0, 0, orc.BLK_INNER)
        W, Z = W0.copy(), Z0.copy()
        d = orc.handle_lambda(ag, W, Z, s, 1, r, ns, term, x)
        Ze = Z0.copy()
        if algo == orc.Q_LAMBDA and 1 != orc.argmax_first(q):
            Ze[:] = 0
        Ze = 0.97 * 0.8 * Ze
        Ze[:, 1] += phi
        if algo == orc.SARSA_LAMBDA:
            na = orc.policy_sample(orc.EGREEDY, qn, x, eps=0.3)
            de = r + 0.97 * qn[na] - q[1]
        else:
            de = r + 0.97 * qn.max() - q[1]
        assert abs(d - de) < 1e-12
        assert np.max(np.abs(Z - Ze)) < 1e-14
        assert np.max(np.abs(W - (W0 + 0.05 * de * Ze))) < 1e-14


def test_pal_hand_computation(orc):
    # control/td/pal.rs:34-60 restated in numpy: persistent advantage learning
    ag = orc.make_agent(algo=orc.PAL, policy=orc.GREEDY, gamma=0.95, lr=0.002, alpha=0.3)
    rng = np.random.default_rng(5)
    for trial in range(20):
        W = rng.normal(size=(36, 3)) * 0.3
        s = np.array([rng.uniform(-1.2, 0.6), rng.uniform(-0.07, 0.07)])
        a = int(rng.integers(0, 3))
        ns, r, term = orc.domain_step(orc.MOUNTAIN_CAR, s, a)
        phi, nphi = orc.fourier_project(0, 5, s), orc.fourier_project(0, 5, ns)
        qs, nqs = phi @ W, nphi @ W
        a_star, na_star = int(np.argmax(qs)), int(np.argmax(nqs))       # distinct values: first == any
        td = r + 0.95 * nqs[a_star] - qs[a]
        al = td - 0.3 * (qs[a_star] - qs[a])
        res = max(al, td - 0.3 * (nqs[na_star] - nqs[a]))
        W_exp = W.copy()
        W_exp[:, a] += 0.002 * (0.3 * res) * phi                        # error sent on = alpha * residual (pal.rs:57)
        W2 = W.copy()
        d = orc.handle(ag, W2, s, a, r, ns, False)
        assert abs(d - res) < 1e-12
        assert np.max(np.abs(W2 - W_exp)) < 1e-14
        W3 = W.copy()
        d = orc.handle(ag, W3, s, a, r, ns, True)                       # terminal: r - Q(s,a)  (pal.rs:39-40)
        assert abs(d - (r - qs[a])) < 1e-12
        assert np.max(np.abs(W3[:, a] - (W[:, a] + 0.002 * 0.3 * d * phi))) < 1e-14
    # PAL's residual never exceeds the plain TD(a*) residual (both corrections are <= 0)  -- the action-gap property
    assert res <= td + 1e-15


def test_greedy_gq_hand_computation(orc):
    # control/td/greedy_gq.rs:73-141 restated in numpy; examples/greedy_gq.rs:25-27 rates (fa_q SGD(0.1), fa_td SGD(0.001))
    ag = orc.make_agent(algo=orc.GREEDY_GQ, order=3, policy=orc.EGREEDY, gamma=0.99, lr=0.1, lr_td=0.001)
    rng = np.random.default_rng(8)
    F = 16
    for trial in range(20):
        W, V = rng.normal(size=(F, 3)) * 0.3, rng.normal(size=(F, 3)) * 0.2
        s = np.array([rng.uniform(-1.2, 0.6), rng.uniform(-0.07, 0.07)])
        a = int(rng.integers(0, 3))
        ns, r, _ = orc.domain_step(orc.MOUNTAIN_CAR, s, a)
        phi, nphi = orc.fourier_project(0, 3, s), orc.fourier_project(0, 3, ns)
        qsa, td_est, nq = phi @ W[:, a], phi @ V[:, a], nphi @ W
        na = int(np.argmax(nq))
        td = r + 0.99 * nq[na] - qsa
        W_exp, V_exp = W.copy(), V.copy()
        W_exp[:, a] += 0.1 * td * phi
        W_exp[:, na] += 0.1 * (-0.99 * td_est) * nphi          # second fa_q update at (s', na)   greedy_gq.rs:113-119
        V_exp[:, a] += 0.001 * (td - td_est) * phi
        W2, V2 = W.copy(), V.copy()
        d = orc.handle_gq(ag, W2, V2, s, a, r, ns, False)
        assert abs(d - td) < 1e-12
        assert np.max(np.abs(W2 - W_exp)) < 1e-14 and np.max(np.abs(V2 - V_exp)) < 1e-14
        # terminal: td_error = r - qsa, no second fa_q update   greedy_gq.rs:79-96
        W3, V3 = W.copy(), V.copy()
        d = orc.handle_gq(ag, W3, V3, s, a, r, ns, True)
        assert abs(d - (r - qsa)) < 1e-12
        W_t, V_t = W.copy(), V.copy()
        W_t[:, a] += 0.1 * d * phi
        V_t[:, a] += 0.001 * (d - td_est) * phi
        assert np.max(np.abs(W3 - W_t)) < 1e-14 and np.max(np.abs(V3 - V_t)) < 1e-14
    # the driver loop keeps fa_td in the run's auxiliary matrix
    run = orc.Run(ag, 3, "f64")
    run.reset()
    run.train(50)
    assert np.abs(run.weights).max() > 0 and np.abs(run.traces).max() > 0


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_optimised_cpu_loop_is_the_same_computation(orc, prec):
    # bench.py's "optimised CPU" baseline (1 projection per step, no heap traffic) must be the reference-pattern loop
    # with the redundancy removed, nothing else: identical states, actions, weights
    kw = dict(policy=orc.EGREEDY, epsilon=0.1, gamma=0.9, lr=0.001, seed=3, max_episode_steps=120)
    a, b = orc.Run(orc.make_agent(**kw), 12, prec), orc.Run(orc.make_agent(**kw), 12, prec)
    a.reset(); b.reset()
    sa = a.train(700)
    sb = b.train_fast(300)
    sb2 = b.train_fast(400)          # split calls: the step counter carries over
    assert np.array_equal(a.state, b.state) and np.array_equal(a.action, b.action)
    assert np.array_equal(a.weights, b.weights)
    assert sa["episodes"] == sb["episodes"] + sb2["episodes"] and sa["env_steps"] == 8400
    assert abs(sa["sum_abs_td_error"] - sb["sum_abs_td_error"] - sb2["sum_abs_td_error"]) < 1e-6 * sa["sum_abs_td_error"]
    with pytest.raises(ValueError):
        orc.Run(orc.make_agent(algo=orc.SARSA), 2, prec).train_fast(1)


def test_td_prediction_hand_computation(orc):
    # prediction/td/td.rs:31-59 and td_lambda.rs:41-78 on a ScalarLFA (fa/linear.rs:201-251), restated in numpy
    rng = np.random.default_rng(12)
    F = 36
    for trace, lam in ((orc.TRACE_ACCUMULATE, 0.6), (orc.TRACE_SATURATE, 0.9), (orc.TRACE_DUTCH, 0.5)):
        ag0 = orc.make_agent(algo=orc.TD, policy=orc.RANDOM, gamma=0.95, lr=0.01)
        agl = orc.make_agent(algo=orc.TD_LAMBDA, policy=orc.RANDOM, gamma=0.95, alpha=0.2, lam=lam, trace=trace)
        for term in (False, True):
            w, z = rng.normal(size=F) * 0.2, rng.normal(size=F) * 0.7
            s = np.array([rng.uniform(-1.2, 0.6), rng.uniform(-0.07, 0.07)])
            ns, r, _ = orc.domain_step(orc.MOUNTAIN_CAR, s, int(rng.integers(0, 3)))
            phi, nphi = orc.fourier_project(0, 5, s), orc.fourier_project(0, 5, ns)
            pred = phi @ w
            assert abs(orc.v_evaluate(ag0, w, s) - pred) < 1e-13
            td = (r - pred) if term else (r + 0.95 * (nphi @ w) - pred)
            # TD(0): w += lr * td * phi(s)
            w2 = w.copy()
            d = orc.handle_td(ag0, w2, None, s, r, ns, term)
            assert abs(d - td) < 1e-12 and np.max(np.abs(w2 - (w + 0.01 * td * phi))) < 1e-14
            # TD(lambda): the trace moves first, then w += td * trace (no learning rate), terminal resets the trace
            rate = 0.95 * lam * ((1 - 0.2) if trace == orc.TRACE_DUTCH else 1.0)
            z_exp = rate * z + phi
            if trace == orc.TRACE_SATURATE:
                z_exp = np.clip(z_exp, -1.0, 1.0)
            w3, z3 = w.copy(), z.copy()
            d = orc.handle_td(agl, w3, z3, s, r, ns, term)
            assert abs(d - td) < 1e-12
            assert np.max(np.abs(w3 - (w + td * z_exp))) < 1e-13
            assert np.max(np.abs(z3 - (0 * z_exp if term else z_exp))) < 1e-14
    # driver loop: Random behaviour policy, one weight column per learner
    run = orc.Run(orc.make_agent(algo=orc.TD, policy=orc.RANDOM, gamma=0.99, lr=0.01, max_episode_steps=50), 5, "f64")
    run.reset()
    st = run.train(120)
    assert run.weights.shape == (5, 36, 1) and np.abs(run.weights).max() > 0 and st["episodes"] == 10
