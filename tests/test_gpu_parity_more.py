"""GPU parity for the other §8 rows: tile coding (bit-exact indices), CartPole / Acrobot learners, shared
weights (synchronous mini-batch rule).  Same tolerances as test_gpu_parity_mc.py."""
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
    s = mid[:, None] + half[:, None] * (2 * rng.random((len(lo), M)) - 1)
    return s.astype(np.float32)


@pytest.mark.parametrize("domain", [0, 1, 2])
@pytest.mark.parametrize("T,B", [(8, 8), (4, 5), (16, 4)])
def test_tile_indices_bit_exact(ra, orc, domain, T, B):
    M = 3000
    s = rand_states(orc, domain, M, 100 * domain + T)
    lo, hi = orc.domain_bounds(domain)
    s[:, 0] = lo.astype(np.float32)              # edge cases: exactly on the bounds
    s[:, 1] = hi.astype(np.float32)
    ag = orc.make_agent(domain=domain, basis=orc.TILE, n_tilings=T, tiles_per_dim=B)
    with ra.Context(domain=domain, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=B, n_envs=M,
                    weight_mode=ra.W_SHARED) as c:
        assert c.F == T * B ** len(lo)
        idx = c.tile_indices(s)
    assert idx.shape == (T, M) and idx.dtype == np.int32
    for m in range(M):
        assert np.array_equal(idx[:, m], orc.tile_indices(ag, s[:, m])), m          # integer path: bit-exact
    assert idx.min() >= 0 and idx.max() < T * B ** len(lo)


def test_tile_q_and_handle_per_env(ra, orc):
    M, T, B = 48, 8, 8
    kw = dict(gamma=0.99, lr=0.1 / T, epsilon=0.1)
    ag = orc.make_agent(domain=1, basis=orc.TILE, n_tilings=T, tiles_per_dim=B, algo=orc.SARSA, policy=orc.EGREEDY, seed=6, **kw)
    rng = np.random.default_rng(0)
    s = rand_states(orc, 1, M, 9, shrink=0.5)
    a = rng.integers(0, 2, M).astype(np.int32)
    with ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=B, algo=ra.SARSA,
                    policy=ra.EPSILON_GREEDY, seed=6, n_envs=M, **kw) as c:
        F = c.F
        Ws = [(rng.normal(size=(F, 2)) * 0.1).astype(np.float32) for _ in range(M)]
        for i in range(M):
            c.set_weights(Ws[i], i)
        assert np.array_equal(c.get_weights(5), Ws[5])
        q = c.q_evaluate(s)
        c.states = np.pad(s, ((0, 0), (0, 0)))
        frm, nxt, rew, term = c.domain_step(a)
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            q32 = orc.q_evaluate(ag, Ws[i], s[:, i], "f32")
            assert np.allclose(q[:, i], q32, rtol=0, atol=1e-6)
            W = Ws[i].copy()
            d = orc.handle(ag, W, frm[:, i], a[i], rew[i], nxt[:, i], term[i], orc.draw(6, i, 0, orc.BLK_INNER), "f32")
            assert abs(td[i] - d) <= 2e-6 * (1 + abs(d))
            assert np.max(np.abs(c.get_weights(i) - W)) <= 1e-6


@pytest.mark.parametrize("domain,basis,algo,policy,kw", [
    (1, "tile", 1, 1, dict(gamma=0.99, lr=0.0125, epsilon=0.1)),          # config C3 in small: CartPole SARSA tiles 8 x 8^4
    (1, "fourier", 0, 1, dict(gamma=0.99, lr=0.01, epsilon=0.1)),         # CartPole QLearning Fourier(1)
    (2, "fourier", 2, 2, dict(gamma=0.99, lr=0.01, alpha=0.5, tau=1.0)),  # Acrobot ExpectedSARSA Fourier(1) Softmax
    (2, "tile", 0, 0, dict(gamma=0.99, lr=0.025)),                        # Acrobot QLearning tiles 4 x 6^4 Greedy
])
def test_train_per_env_vs_oracle_f32(ra, orc, domain, basis, algo, policy, kw):
    N, K = 96, 60
    if basis == "tile":
        T, B = (8, 8) if domain == 1 else (4, 6)
        okw = dict(basis=orc.TILE, n_tilings=T, tiles_per_dim=B)
        dkw = dict(basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=B)
    else:
        okw, dkw = dict(basis=orc.FOURIER, order=1), dict(basis=ra.FOURIER, order=1)
    ag = orc.make_agent(domain=domain, algo=algo, policy=policy, seed=13, max_episode_steps=40, **okw, **kw)
    run = orc.Run(ag, N, "f32")
    run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, algo=algo, policy=policy, seed=13, max_episode_steps=40, n_envs=N, **dkw, **kw) as c:
        c.reset()
        st = c.train(K)
        tol = 2e-3 if domain == 2 else 1e-5            # Acrobot: chaotic RK4 (dt 0.2) amplifies sincos ulps over 60 steps
        same = np.all(np.abs(c.states.T - run.state) <= tol * (1 + np.abs(run.state)), axis=1) & (c.actions == run.action)
        assert same.mean() >= (0.85 if policy == 2 or domain == 2 else 0.95), same.mean()
        for i in np.flatnonzero(same)[:10]:
            assert np.max(np.abs(c.get_weights(i) - run.weights[i])) <= (5e-4 if domain == 2 else 5e-6)
        assert abs(st["episodes"] - ost["episodes"]) <= max(2, 0.05 * ost["episodes"])
        assert st["env_steps"] == N * K


@pytest.mark.parametrize("basis", ["fourier", "tile"])
def test_train_shared_weights_vs_oracle(ra, orc, basis):
    # shared approximator: W_{t+1} = W_t + lr * sum_i e_i phi(s_i) x onehot(a_i), all errors against W_t
    N, K = 600, 40                       # 600 envs = 3 thread blocks, the last one partially filled
    if basis == "tile":
        domain, okw, dkw, kw = 1, dict(basis=orc.TILE, n_tilings=8, tiles_per_dim=8), \
            dict(basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8), dict(gamma=0.99, lr=0.0125 / 50, epsilon=0.1)
        algo, policy = 1, 1
    else:
        domain, okw, dkw, kw = 0, dict(order=5), dict(order=5), dict(gamma=0.9, lr=0.001 / 50, epsilon=0.1)
        algo, policy = 0, 1
    ag = orc.make_agent(domain=domain, algo=algo, policy=policy, shared_w=True, seed=17, max_episode_steps=25, **okw, **kw)
    run = orc.Run(ag, N, "f32")
    run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, algo=algo, policy=policy, weight_mode=ra.W_SHARED, seed=17, max_episode_steps=25,
                    n_envs=N, **dkw, **kw) as c:
        c.reset()
        st = c.train(K)
        Wd, Wo = c.get_weights(), run.weights
        assert np.max(np.abs(Wo)) > 1e-5
        assert np.max(np.abs(Wd - Wo)) <= 2e-5 * max(1.0, np.max(np.abs(Wo))) + 1e-7
        same = np.all(np.abs(c.states.T - run.state) <= 1e-5, axis=1) & (c.actions == run.action)
        assert same.mean() >= 0.95, same.mean()
        assert abs(st["episodes"] - ost["episodes"]) <= max(2, 0.02 * ost["episodes"])
        assert abs(st["sum_abs_td_error"] - ost["sum_abs_td_error"]) <= 1e-3 * ost["sum_abs_td_error"]


def test_shared_weights_dense_is_reproducible(ra):
    # the dense delta reduction uses fixed-order block partials (no atomics): same seed => bitwise same W
    kw = dict(n_envs=5000, weight_mode=ra.W_SHARED, policy=1, epsilon=0.1, lr=1e-6, seed=4, max_episode_steps=50)
    out = []
    for _ in range(2):
        with ra.Context(**kw) as c:
            c.reset()
            c.train(60)
            out.append((c.get_weights(), c.states, c.actions))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])


def test_shared_weights_n1_equals_reference_rule(ra):
    # N = 1: the mini-batch rule collapses to the reference's one-update-at-a-time rule (per-env mode, N = 1)
    kw = dict(n_envs=1, policy=1, epsilon=0.1, seed=9, max_episode_steps=100, steps_per_launch=1)
    with ra.Context(weight_mode=ra.W_SHARED, **kw) as a, ra.Context(weight_mode=ra.W_PER_ENV, **kw) as b:
        a.reset(), b.reset()
        a.train(300), b.train(300)
        assert np.allclose(a.get_weights(), b.get_weights(0), rtol=0, atol=1e-7)
        assert np.allclose(a.states, b.states, atol=1e-6)


def test_shared_handle_minibatch(ra, orc):
    # rsrl_hip_handle in shared mode: all M errors against the same W_t, then one summed update
    M = 64
    ag = orc.make_agent(policy=orc.GREEDY, shared_w=True, gamma=0.9, lr=0.01)
    rng = np.random.default_rng(3)
    W0 = (rng.normal(size=(36, 3)) * 0.1).astype(np.float32)
    s = rand_states(orc, 0, M, 5)
    a = rng.integers(0, 3, M).astype(np.int32)
    with ra.Context(n_envs=M, weight_mode=ra.W_SHARED, policy=0, gamma=0.9, lr=0.01) as c:
        c.set_weights(W0)
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        td = c.handle(frm, a, rew, nxt, term)
        Wd = c.get_weights()
    W = W0.astype(np.float64)
    dW = np.zeros_like(W)
    for i in range(M):
        Wi = W.copy()
        d = orc.handle(ag, Wi, frm[:, i], a[i], rew[i], nxt[:, i], term[i])
        dW += Wi - W
        assert abs(td[i] - d) <= 2e-5 * (1 + abs(d))
    assert np.max(np.abs(Wd - (W + dW))) <= 2e-6


@pytest.mark.parametrize("domain,order", [(0, 7), (0, 6), (1, 2), (2, 3)])
def test_generic_fourier_orders(ra, orc, domain, order):
    # Fourier orders without a specialised kernel go through the generic on-the-fly model: same arithmetic
    N, K = 40, 30
    D = 2 if domain == 0 else 4
    F = (order + 1) ** D
    rng = np.random.default_rng(order)
    s = rand_states(orc, domain, N, 40 + order, shrink=0.6)
    kw = dict(gamma=0.95, lr=0.01, epsilon=0.1)
    ag = orc.make_agent(domain=domain, order=order, policy=orc.EGREEDY, seed=31, max_episode_steps=20, **kw)
    with ra.Context(domain=domain, order=order, policy=1, seed=31, max_episode_steps=20, n_envs=N, **kw) as c:
        assert c.F == F
        phi = c.project(s)
        for m in range(0, N, 5):
            assert np.max(np.abs(phi[:, m] - orc.fourier_project(domain, order, s[:, m], "f32"))) <= 2e-6
            assert np.max(np.abs(phi[:, m] - orc.fourier_project(domain, order, s[:, m], "f64"))) <= 6e-6
        A = c.A
        Ws = [(rng.normal(size=(F, A)) * 0.1).astype(np.float32) for _ in range(N)]
        for i in range(N):
            c.set_weights(Ws[i], i)
        q = c.q_evaluate(s)
        for i in range(0, N, 3):
            q32 = orc.q_evaluate(ag, Ws[i], s[:, i], "f32")
            assert np.allclose(q[:, i], q32, rtol=0, atol=1e-5 * (1 + np.abs(q32).max()))
        for i in range(N):
            c.set_weights(np.zeros((F, A), dtype=np.float32), i)
        run = orc.Run(ag, N, "f32")
        run.reset()
        run.train(K)
        c.reset()
        st = c.train(K)
        tol = 2e-3 if domain == 2 else 1e-5
        same = np.all(np.abs(c.states.T - run.state) <= tol * (1 + np.abs(run.state)), axis=1) & (c.actions == run.action)
        assert same.mean() >= 0.85, same.mean()
        for i in np.flatnonzero(same)[:5]:
            assert np.max(np.abs(c.get_weights(i) - run.weights[i])) <= (5e-4 if domain == 2 else 5e-6)
        n, _ = c.rollout_greedy(50)
        assert n.shape == (N,) and st["env_steps"] == N * K
    with pytest.raises(ra.RsrlHipError):
        ra.Context(domain=domain, order=order, n_envs=4, weight_mode=ra.W_SHARED)
    with pytest.raises(ra.RsrlHipError):
        ra.Context(domain=domain, order=8, n_envs=4)


def test_checkpoint_roundtrip_and_resume(ra, tmp_path):
    # save -> load into a fresh ctx -> continue: identical to never having stopped (weights + step counter restored;
    # env state is re-created by reset, so compare a run that also resets at the same point)
    kw = dict(n_envs=300, policy=1, epsilon=0.1, seed=12, max_episode_steps=40)
    path = tmp_path / "w.rsrlw"
    with ra.Context(**kw) as a:
        a.reset(); a.train(50)
        a.save_weights(path)
        wa = [a.get_weights(i) for i in (0, 150, 299)]
        a.reset(); a.train(30)
        ref = (a.states.copy(), [a.get_weights(i) for i in (0, 150, 299)])
    raw = open(path, "rb").read()
    assert raw[:8] == b"RSRLHIPW" and len(raw) == 72 + 300 * 36 * 3 * 4
    import struct                                      # the documented byte layout (include/rsrl_hip.h), field by field
    ver, *f11 = struct.unpack_from("<I11i", raw, 8)
    n_learners, step_count = struct.unpack_from("<qQ", raw, 56)
    assert ver == 2 and f11 == [0, 0, 5, 8, 8, 0, 36, 3, 0, 0, 0] and (n_learners, step_count) == (300, 50)
    w0 = np.frombuffer(raw, dtype="<f4", count=108, offset=72).reshape(36, 3)
    assert np.array_equal(w0, wa[0])
    with ra.Context(**kw) as b:
        b.load_weights(path)
        assert b.step_count == 50
        for w, i in zip(wa, (0, 150, 299)):
            assert np.array_equal(b.get_weights(i), w)
        b.reset(); b.train(30)
        assert np.array_equal(b.states, ref[0])
        for w, i in zip(ref[1], (0, 150, 299)):
            assert np.array_equal(b.get_weights(i), w)
    with ra.Context(n_envs=299, policy=1) as c:
        with pytest.raises(ra.RsrlHipError):
            c.load_weights(path)                      # different configuration
    with ra.Context(algo=ra.SARSA, **kw) as c:
        with pytest.raises(ra.RsrlHipError):
            c.load_weights(path)                      # different agent (the header carries algo and weight dtype)
    # a truncated file is refused before anything is touched: the ctx keeps its weights and its step counter
    cut = tmp_path / "cut.rsrlw"
    cut.write_bytes(raw[:72 + 150 * 432 + 100])
    with ra.Context(**kw) as c:
        c.reset(); c.train(20)
        before = [c.get_weights(i) for i in (0, 149, 150, 299)]
        with pytest.raises(ra.RsrlHipError) as ei:
            c.load_weights(cut)
        assert "truncated" in str(ei.value)
        assert c.step_count == 20
        for w, i in zip(before, (0, 149, 150, 299)):
            assert np.array_equal(c.get_weights(i), w)
        c.train(5)                                     # and still works
    # GreedyGQ: the second approximator (fa_td) travels with the checkpoint -> the resumed run is bit-identical
    gkw = dict(n_envs=64, algo=ra.GREEDY_GQ, policy=1, epsilon=0.1, seed=3, max_episode_steps=40, lr=0.05, lr_td=0.002)
    gpath = tmp_path / "gq.rsrlw"
    with ra.Context(**gkw) as a:
        a.reset(); a.train(60)
        a.save_weights(gpath)
        v7 = a.get_td_weights(7)
        a.reset(); a.train(25)
        gref = (a.states.copy(), a.get_weights(7), a.get_td_weights(7))
    assert len(open(gpath, "rb").read()) == 72 + 2 * 64 * 432 and np.abs(v7).max() > 0
    with ra.Context(**gkw) as b:
        b.load_weights(gpath)
        assert np.array_equal(b.get_td_weights(7), v7)
        b.reset(); b.train(25)
        assert np.array_equal(b.states, gref[0]) and np.array_equal(b.get_weights(7), gref[1]) and np.array_equal(b.get_td_weights(7), gref[2])
    with ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=4, tiles_per_dim=4, weight_mode=ra.W_SHARED, n_envs=8) as t:
        t.set_weights(np.arange(4 * 256 * 2, dtype=np.float32).reshape(1024, 2))
        t.save_weights(tmp_path / "t.rsrlw")
        t.set_weights(np.zeros((1024, 2), dtype=np.float32))
        t.load_weights(tmp_path / "t.rsrlw")
        assert np.array_equal(t.get_weights(), np.arange(2048, dtype=np.float32).reshape(1024, 2))


@pytest.mark.parametrize("name,kw,bitwise", [
    ("k1", dict(n_envs=5000, policy=1, epsilon=0.1, seed=2, max_episode_steps=60, steps_per_launch=1), True),
    ("shared-dense", dict(n_envs=5000, policy=1, epsilon=0.1, seed=2, max_episode_steps=60, weight_mode=1, lr=1e-5), True),
    ("shared-tile", dict(domain=1, basis=1, algo=1, n_envs=5000, policy=1, epsilon=0.1, seed=2, max_episode_steps=60, weight_mode=1, lr=1e-4), False),
])
def test_graph_replay_equals_plain_launches(ra, monkeypatch, name, kw, bitwise):
    # the launch-bound loops (one batch-step per launch) are replayed as a captured hipGraph of 32 steps whose nodes read
    # the step counter from the device; RSRL_NO_GRAPH=1 keeps plain launches.  Same results either way, including across
    # a change of epsilon between replays (the nodes read the policy parameters from device memory: no re-capture) and chunk sizes that are not multiples of 32.
    def run(no_graph):
        if no_graph:
            monkeypatch.setenv("RSRL_NO_GRAPH", "1")
        else:
            monkeypatch.delenv("RSRL_NO_GRAPH", raising=False)
        with ra.Context(**kw) as c:
            c.reset()
            c.train(100, want_stats=False)
            c.train(7, want_stats=False)
            c.set_epsilon(0.05)
            c.train(77, want_stats=False)
            st = c.train(40)                   # statistics requested: plain launches
            assert c.step_count == 224
            return c.get_weights(3), c.states, c.actions, st["episodes"]
    a, b = run(False), run(True)
    if bitwise:
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    else:                                      # f32 atomics: the order of the adds is not fixed
        assert np.allclose(a[0], b[0], rtol=0, atol=1e-6) and (np.abs(a[1] - b[1]) <= 1e-5).mean() > 0.98


def test_single_step_layout_is_invisible_through_the_abi(ra, orc, tmp_path, monkeypatch):
    # a ctx created with steps_per_launch = 1 keeps W learner-major (k_step_reg_lm writes back only the touched column);
    # nothing of that may show through the ABI: weights in/out, Q, handle, checkpoints, checksums, training results
    kw = dict(n_envs=777, policy=1, epsilon=0.1, seed=5, max_episode_steps=70)
    rng = np.random.default_rng(3)
    Ws = {i: (rng.normal(size=(36, 3)) * 0.1).astype(np.float32) for i in (0, 1, 63, 64, 500, 776)}
    s = (np.array([[-0.5], [0.0]]) + rng.normal(size=(2, 777)) * np.array([[0.3], [0.02]])).astype(np.float32)
    with ra.Context(steps_per_launch=1, **kw) as lm, ra.Context(**kw) as fm:
        for i, W in Ws.items():
            lm.set_weights(W, i); fm.set_weights(W, i)
            assert np.array_equal(lm.get_weights(i), W)
        assert np.array_equal(lm.q_evaluate(s), fm.q_evaluate(s))
        assert lm.checksum() == fm.checksum()
        lm.reset(); fm.reset()
        lm.train(150); fm.train(150)                      # 150 single-step launches (graph + plain) vs one fused launch
        assert np.array_equal(lm.states, fm.states) and np.array_equal(lm.actions, fm.actions)
        for i in (0, 63, 64, 776):
            assert np.array_equal(lm.get_weights(i), fm.get_weights(i))
        assert lm.checksum() == fm.checksum()
        a = lm.actions.copy()
        frm, nxt, rew, term = lm.domain_step(a)
        fm.states = frm
        fm.domain_step(a)
        td1, td2 = lm.handle(frm, a, rew, nxt, term), fm.handle(frm, a, rew, nxt, term)
        assert np.array_equal(td1, td2) and np.array_equal(lm.get_weights(64), fm.get_weights(64))
        path = tmp_path / "w.rsrl"
        lm.save_weights(path)
        fm.set_weights_all(np.zeros((36, 3), np.float32))
        fm.load_weights(path)
        assert np.array_equal(lm.get_weights(500), fm.get_weights(500)) and lm.checksum()[0] == fm.checksum()[0]
    # the feature-major single-step kernel stays available and gives the same bits
    monkeypatch.setenv("RSRL_K1_FEATURE_MAJOR", "1")
    with ra.Context(steps_per_launch=1, **kw) as c1:
        monkeypatch.delenv("RSRL_K1_FEATURE_MAJOR")
        with ra.Context(steps_per_launch=1, **kw) as c2:
            c1.reset(); c2.reset()
            c1.train(90); c2.train(90)
            assert np.array_equal(c1.get_weights(5), c2.get_weights(5)) and np.array_equal(c1.states, c2.states)
            assert c1.checksum() == c2.checksum()


@pytest.mark.parametrize("name,okw,dkw,K", [
    ("cartpole-sarsa-tiles", dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=1, policy=1, epsilon=0.1, gamma=0.99, lr=0.0125),
     dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=1, policy=1, epsilon=0.1, gamma=0.99, lr=0.0125), 5000),
    ("acrobot-esarsa-softmax", dict(domain=2, order=1, algo=2, policy=2, tau=1.0, gamma=0.99, lr=0.01, alpha=0.5),
     dict(domain=2, order=1, algo=2, policy=2, tau=1.0, gamma=0.99, lr=0.01, alpha=0.5), 3000),
])
def test_long_run_population_statistics_other_configs(ra, orc, name, okw, dkw, K):
    # the other agents / bases / policies over thousands of steps: individual trajectories part ways (fp32 vs f64, chaotic
    # dynamics), the learner population must not -- episodes finished, TD error mass and weight magnitude within 2-3 %
    N = 128
    ag = orc.make_agent(seed=21, max_episode_steps=200, **okw)
    run = orc.Run(ag, N, "f64")
    run.reset()
    ost = run.train(K)
    with ra.Context(n_envs=N, seed=21, max_episode_steps=200, **dkw) as c:
        c.reset()
        st = c.train(K)
        Wd = np.stack([c.get_weights(i) for i in range(N)])
        assert abs(st["episodes"] - ost["episodes"]) <= 0.03 * ost["episodes"] + 3, (st["episodes"], ost["episodes"])
        assert abs(st["sum_abs_td_error"] - ost["sum_abs_td_error"]) <= 0.03 * ost["sum_abs_td_error"]
        assert abs(np.abs(Wd).mean() - np.abs(run.weights).mean()) <= 0.03 * np.abs(run.weights).mean()
        assert abs(st["sum_reward"] - ost["sum_reward"]) <= 0.03 * abs(ost["sum_reward"]) + 3


def test_launch_coalescing_is_invisible(ra, monkeypatch):
    # rsrl_hip_train holds back short calls that arrive while the stream is busy and launches them fuse-depth at a time; every
    # other entry point flushes first.  Same results bit for bit, same step counter, parameter changes land where they were made.
    kw = dict(n_envs=5000, policy=1, epsilon=0.2, seed=3, max_episode_steps=50)

    def drive(c):
        c.reset()
        for _ in range(200):
            c.train(7, want_stats=False)
        assert c.step_count == 1400                 # accepted steps count, launched or not
        c.set_epsilon(0.05)                         # flushes the pending steps with the OLD epsilon first
        for _ in range(50):
            c.train(3, want_stats=False)
        st = c.train(10)                            # statistics: immediate launch after a flush
        return c.checksum(), c.states, c.actions, c.get_weights(4999), st, c.step_count

    with ra.Context(**kw) as a:
        got = drive(a)
    monkeypatch.setenv("RSRL_NO_COALESCE", "1")
    with ra.Context(**kw) as b:
        ref = drive(b)
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    assert got[4] == ref[4] and got[5] == ref[5] == 1560
