"""CPU tests of the oracle's round-4 additions (test infrastructure checked against the reference's own lines):
  * the drivers' epsilon schedule, `agent.policy.epsilon *= 0.995` once per episode of a learner (rsrl/examples/sarsa_lambda.rs:48-75,
    :68; the pub field policies/epsilon_greedy.rs:19) -- against an independent transcription of that loop;
  * Domain::rollout with an arbitrary policy closure (rsrl_domains/src/lib.rs:448-479)."""
import numpy as np
import pytest

EX = dict(domain=0, order=5, algo=3, policy=1, trace=1, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)      # examples/sarsa_lambda.rs:19-45


def test_schedule_is_the_reference_drivers(orc):
    # ONE learner, f64: an independent transcription of examples/sarsa_lambda.rs:48-75 -- episode loop, `epsilon *= 0.995` after the
    # episode's last sample (:68) -- driven by the oracle's single-transition functions, against the oracle's vectorised loop
    ag = orc.make_agent(seed=11, max_episode_steps=30, epsilon_decay=0.995, **EX)
    run = orc.Run(ag, 1, "f64")
    run.reset()
    K = 400
    F, A = orc.n_features(ag), 3
    W, Z = np.zeros((F, A)), np.zeros((F, A))
    eps, t = 0.2, 0
    s = orc.domain_reset(0)
    a = orc.policy_sample(orc.EGREEDY, orc.q_evaluate(ag, W, s), orc.draw(11, 0, 0, orc.BLK_INIT), eps=eps)
    episodes, ep_len, eps_hist = 0, 0, []
    for t in range(K):
        ns, r, term = orc.domain_step(0, s, a)
        agl = orc.make_agent(seed=11, max_episode_steps=30, **{**EX, "epsilon": eps})        # agent.policy is the shared object: SARSALambda's own draw too
        orc.handle_lambda(agl, W, Z, s, a, r, ns, term, orc.draw(11, 0, t, orc.BLK_INNER))
        ep_len += 1
        done = term or ep_len >= 30
        if done:
            eps = eps * 0.995                                     # :68
            episodes += 1; ep_len = 0
            ns = orc.domain_reset(0)
        a = orc.policy_sample(orc.EGREEDY, orc.q_evaluate(ag, W, ns), orc.draw(11, 0, t, orc.BLK_RESET if done else orc.BLK_STEP), eps=eps)
        s = ns
        eps_hist.append(eps)
    st = run.train(K)
    assert st["episodes"] == episodes >= 10
    assert run.eps[0] == eps                                      # the very same f64 products
    assert np.array_equal(run.state[0], s) and run.action[0] == a
    assert np.max(np.abs(run.weights[0] - W)) == 0.0 and np.max(np.abs(run.traces[0] - Z)) == 0.0
    e = 0.2
    for _ in range(episodes):
        e *= 0.995
    assert e == eps


def test_schedule_properties(orc):
    N, K = 24, 600
    ag = orc.make_agent(seed=2, max_episode_steps=12, epsilon_decay=0.9, epsilon_min=0.04, **EX)
    run = orc.Run(ag, N, "f64"); run.reset()
    assert np.all(run.eps == 0.2)
    run.train(K)
    assert run.eps.min() == 0.04 and run.eps.max() <= 0.2 * 0.9 ** 3          # floored; every learner finished >= 50 episodes
    # no schedule: the per-learner field never moves; set_epsilon writes every learner's field (and the agent's shared policy object)
    plain = orc.Run(orc.make_agent(seed=2, max_episode_steps=12, **EX), N, "f64"); plain.reset(); plain.train(50)
    assert np.all(plain.eps == 0.2)
    plain.set_epsilon(0.5)
    assert np.all(plain.eps == 0.5)
    # decay 1.0 == no schedule, bit for bit
    one = orc.Run(orc.make_agent(seed=2, max_episode_steps=12, epsilon_decay=1.0, **EX), N, "f32d"); one.reset(); one.train(100)
    ref = orc.Run(orc.make_agent(seed=2, max_episode_steps=12, **EX), N, "f32d"); ref.reset(); ref.train(100)
    assert np.array_equal(one.weights, ref.weights) and np.array_equal(one.state, ref.state)
    # the float instantiations keep the field in fp32 (the device's): one rounding per episode away from the f64 schedule
    f32 = orc.Run(orc.make_agent(seed=2, max_episode_steps=12, epsilon_decay=0.995, **EX), 4, "f32d"); f32.reset(); f32.train(240)
    f64 = orc.Run(orc.make_agent(seed=2, max_episode_steps=12, epsilon_decay=0.995, **EX), 4, "f64"); f64.reset(); f64.train(240)
    assert f32.eps.dtype == np.float32 and np.max(np.abs(f32.eps - f64.eps)) <= 25 * 6e-8 * 0.2


@pytest.mark.parametrize("algo,apol", [(0, None), (1, None), (2, None), (1, 0)])
def test_schedule_device_order_loop_follows_the_reference_order_loop(orc, algo, apol):
    kw = dict(algo=algo, policy=1, gamma=0.9, lr=0.001, alpha=0.7, epsilon=0.3, seed=21, max_episode_steps=20, epsilon_decay=0.97, epsilon_min=0.01)
    if apol is not None:
        kw["agent_policy"] = apol
    ag = orc.make_agent(**kw)
    N, K = 32, 500
    ref = orc.Run(ag, N, "f64"); ref.reset(); st = ref.train(K)
    dev = orc.Run(ag, N, "f64"); dev.reset(); dev.train_dev(200); dev.train_dev(K - 200)
    assert np.array_equal(ref.eps, dev.eps) and st["episodes"] >= 25 * N
    same = np.all(np.abs(ref.state - dev.state) <= 1e-9, axis=1) & (ref.action == dev.action)
    assert same.mean() >= 0.95 and np.max(np.abs(ref.weights[same] - dev.weights[same])) <= 1e-10
    with pytest.raises(ValueError):
        dev.train_fast(1)                                         # (the optimised-CPU loop has no schedule)


def test_rollout_policy(orc):
    N, L = 64, 150
    ag = orc.make_agent(policy=orc.EGREEDY, seed=4, lr=0.002, max_episode_steps=100)
    run = orc.Run(ag, N, "f64"); run.reset(); run.train(500)
    n_g, tot_g = run.rollout_greedy(L)
    # epsilon = 0: Greedy::sample == Greedy::mode unless two action values are within 1e-7 of each other (utils.rs:6-21 vs core.rs:96-105)
    n0, tot0, a0 = run.rollout_policy(orc.EGREEDY, L, epsilon=0.0)
    assert (n0 == n_g).mean() >= 0.98 and np.array_equal(tot0[n0 == n_g], tot_g[n0 == n_g])
    # reproducible per call number, another stream for another call
    n1, _, a1 = run.rollout_policy(orc.EGREEDY, L, epsilon=0.5, call=0)
    n2, _, a2 = run.rollout_policy(orc.EGREEDY, L, epsilon=0.5, call=0)
    n3, _, a3 = run.rollout_policy(orc.EGREEDY, L, epsilon=0.5, call=1)
    assert np.array_equal(a1, a2) and np.array_equal(n1, n2) and not np.array_equal(a1, a3)
    # Random: uniform actions, whatever the action values (random.rs:43-45)
    _, _, ar = run.rollout_policy(orc.RANDOM, L)
    fr = np.bincount(ar[:20].ravel(), minlength=3) / ar[:20].size
    assert np.abs(fr - 1 / 3).max() < 0.05
    # n_states = 1 + transitions <= limit; total reward = -(transitions) unless the goal was reached (MountainCar: 0 on the last one)
    assert n1.max() <= L and np.all(n1 >= 2)
    # limit 1: no transition is kept (lib.rs:457-479)
    n, tot, acts = run.rollout_policy(orc.SOFTMAX, 1, tau=0.5)
    assert np.all(n == 1) and np.all(tot == 0) and acts.shape == (0, N)
    with pytest.raises(ValueError):
        run.rollout_policy(orc.EGREEDY, 0)
    with pytest.raises(ValueError):
        run.rollout_policy(9, 10)


def test_constant_division_is_the_ieee_quotient(orc, tmp_path):
    # The HIP path divides by compile-time constants through f64 -- (float)((double)x * (1.0 / (double)d)), three instructions instead of
    # the ~11 of an IEEE fp32 division (rsrl_amd/csrc/device_core.hpp div_const) -- while the oracle keeps the reference's `/`
    # (ode.rs:36 `/ 6.0`, cart_pole.rs:60 `/ TOTAL_MASS`, the tile coder's (s - lo) / (hi - lo)).  They are the same function: shown here
    # by trying EVERY fp32 input (zeros of both signs, denormals, infinities, NaNs) for every divisor on the path.
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    exe = str(tmp_path / "check_constdiv")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-pthread", "-o", exe,
                           os.path.join(here, "check_constdiv.c"), "-lm"])
    divisors = {np.float32(6.0), np.float32(1.0) + np.float32(0.1)}                    # RK4's / 6.0; TOTAL_MASS = 1.0 + 0.1 (consts.rs:4-7)
    for dom in (0, 1, 2):
        lo, hi = orc.domain_bounds(dom)
        for a, b in zip(lo.astype(np.float32), hi.astype(np.float32)):
            divisors.add(np.float32(b - a))                                            # hi - lo in fp32, as the device's constexpr evaluates it
    assert np.float32(1.1) in divisors and len(divisors) >= 10
    n = min(16, len(os.sched_getaffinity(0)))
    args = [exe, "0", "1", str(n), "0", "3.5e38"] + [f"{float(d):.9g}" for d in sorted(divisors)]
    out = subprocess.run(args, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout
    rows = [ln for ln in out.stdout.splitlines() if ln.startswith("d=")]
    assert len(rows) == len(divisors) and all("4294967296 inputs, 0 mismatches" in ln for ln in rows), out.stdout
    # the checker does find a sequence that is NOT the quotient: Markstein's fp32 correction step fails where its residual underflows
    bad = subprocess.run([exe, "1", "1", str(n), "0", "1e-36", "1.10000002"], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1 and "mismatches" in bad.stdout and " 0 mismatches" not in bad.stdout


# The independent leg of the chain for the round-4 families: the device is compared BITWISE with the oracle's f32d
# instantiation (tests/test_gpu_round4.py), which restates the device's own polynomials and evaluation order -- so f32d itself has to be
# held to the reference's arithmetic: the f64 run of the same agent, same seeds, within a tolerance.
R4_FAMILIES = [
    ("greedy_gq, CartPole tiles", dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=6, policy=1, gamma=0.99, lr=0.0125, lr_td=0.001, epsilon=0.1), 0.95, 3e-8),
    ("greedy_gq, MountainCar Fourier(6)", dict(domain=0, order=6, algo=6, policy=1, gamma=0.99, lr=0.05, lr_td=0.001, epsilon=0.1), 0.95, 1e-5),
    ("q_sigma n=3, CartPole tiles", dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=9, policy=1, gamma=0.95, lr=0.0125, alpha=0.5, sigma=0.5, n_steps=3,
                                         epsilon=0.2), 0.95, 5e-9),
    ("q_sigma n=4, MountainCar Fourier(7)", dict(domain=0, order=7, algo=9, policy=1, gamma=0.9, lr=0.01, alpha=0.5, sigma=0.0, n_steps=4, epsilon=0.2), 0.95, 1e-6),
    ("td, MountainCar Fourier(6)", dict(domain=0, order=6, algo=7, policy=3, gamma=0.9, lr=0.01), 1.0, 2e-6),
    ("td, CartPole Fourier(2)", dict(domain=1, order=2, algo=7, policy=3, gamma=0.9, lr=0.01), 1.0, 2e-7),
    ("sarsa + epsilon schedule", dict(domain=0, order=3, algo=1, policy=1, gamma=0.9, lr=0.002, epsilon=0.4, epsilon_decay=0.97, epsilon_min=0.01), 0.95, 2e-7),
]


@pytest.mark.parametrize("name,kw,min_same,tol", R4_FAMILIES, ids=[c[0] for c in R4_FAMILIES])
def test_round4_families_f32d_follows_the_f64_reference_run(orc, name, kw, min_same, tol):
    N, K = 48, 160
    ag = orc.make_agent(seed=17, max_episode_steps=25, **kw)
    d = orc.Run(ag, N, "f32d"); d.reset(); sd = d.train(K)
    r = orc.Run(ag, N, "f64"); r.reset(); sr = r.train(K)
    same = np.all(np.abs(r.state - d.state) <= 1e-4 * (1 + np.abs(r.state)), axis=1) & (r.action == d.action)
    assert same.mean() >= min_same, same.mean()                  # (fp32 argmax flips move a learner to another trajectory: inherent)
    scale = max(1.0, float(np.abs(r.weights[same]).max()))
    assert np.max(np.abs(r.weights[same] - d.weights[same])) <= tol * scale
    if kw["algo"] == 6:
        assert np.max(np.abs(r.traces[same] - d.traces[same])) <= tol * max(1.0, float(np.abs(r.traces[same]).max()))
    assert abs(sd["episodes"] - sr["episodes"]) <= max(2, int((1 - same.mean()) * N * K / 10))
    assert np.abs(r.weights).max() > 0
    if "epsilon_decay" in kw:
        assert np.max(np.abs(r.eps[same] - d.eps[same])) <= 30 * 6e-8 * 0.4
