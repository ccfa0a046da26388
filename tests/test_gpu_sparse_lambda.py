"""Eligibility traces over ONE SHARED tile-coded table (VERDICT r4 missing #1): SARSALambda / QLambda with weight_mode = SHARED, basis = TILE_CODING.
The reference's Trace<B, R> is generic over its buffer and ships a sparse one (traces.rs:5-12, params/sparse.rs:13-97): every learner keeps a sparse
trace (<= 512 entries, as one sub-list of 512 / T per tiling), the table is updated by the synchronous mini-batch rule through exact fixed-point sums
(rsrl_amd/csrc/kernels_sparse_lambda.hpp: the step kernel of the one-step shared-table agents + the trace update inside the LDS scatter kernel).
Bitwise against the oracle's restatement (f32d), tolerance against the same rule in f64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


CASES = [
    # name, N, K, device kwargs
    ("cartpole_sarsa_accumulate", 300, 200, dict(domain=1, n_tilings=8, tiles_per_dim=8, algo=3, policy=1, epsilon=0.1, gamma=0.99, lam=0.9, trace=0, max_episode_steps=60)),
    ("cartpole_q_saturate", 300, 200, dict(domain=1, n_tilings=8, tiles_per_dim=8, algo=4, policy=1, epsilon=0.2, gamma=0.99, lam=0.8, trace=1, max_episode_steps=60)),
    ("mountaincar_sarsa_dutch_evicting", 64, 900, dict(domain=0, n_tilings=8, tiles_per_dim=8, algo=3, policy=1, epsilon=0.3, gamma=0.99, lam=0.97, trace=2, max_episode_steps=0)),
    ("acrobot_q_accumulate_softmax", 96, 120, dict(domain=2, n_tilings=4, tiles_per_dim=6, algo=4, policy=2, tau=0.5, gamma=0.95, lam=0.7, trace=0, max_episode_steps=50)),
    # a tiling's slice of 20 000 entries (160 KB as 64-bit accumulators: beyond the LDS): the scatter kernel adds its terms with device atomics instead
    ("cartpole_q_slice_beyond_lds", 128, 80, dict(domain=1, n_tilings=4, tiles_per_dim=10, algo=4, policy=1, epsilon=0.2, gamma=0.99, lam=0.8, trace=0, max_episode_steps=40)),
    # learner counts that leave a wave's last 16-lane groups without a learner (four learners per wave), 16 tilings (two registers per sub-list)
    ("cartpole_sarsa_ragged_301", 301, 150, dict(domain=1, n_tilings=8, tiles_per_dim=8, algo=3, policy=1, epsilon=0.1, gamma=0.99, lam=0.9, trace=0, max_episode_steps=60)),
    ("mountaincar_sarsa_16_tilings_77_evicting", 77, 500, dict(domain=0, n_tilings=16, tiles_per_dim=8, algo=3, policy=1, epsilon=0.3, gamma=0.99, lam=0.95, trace=1, max_episode_steps=0)),
    ("cartpole_sarsa_4096", 4096, 60, dict(domain=1, n_tilings=8, tiles_per_dim=8, algo=3, policy=1, epsilon=0.1, gamma=0.99, lam=0.9, trace=0, max_episode_steps=40)),
]


@pytest.mark.parametrize("name,N,K,kw", CASES, ids=[c[0] for c in CASES])
def test_sparse_traces_bitwise_and_f64(ra, orc, name, N, K, kw):
    alpha = 0.1 / kw["n_tilings"] / N
    okw = dict(kw, basis=orc.TILE, shared_w=True, seed=7, alpha=alpha)
    ag = orc.make_agent(**okw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    ost = run.train_sparse_lambda(K)
    r64 = orc.Run(ag, N, "f64")
    r64.reset()
    st64 = r64.train_sparse_lambda(K)
    with ra.Context(basis=ra.TILE_CODING, weight_mode=ra.W_SHARED, seed=7, alpha=alpha, n_envs=N, **kw) as c:
        c.reset()
        sat0 = c.fx_saturations()                                   # (a counter of the device / process, not of the ctx)
        st = c.train(K // 2)
        st2 = c.train(K - K // 2)                                   # any split into calls
        W = c.get_weights()
        assert np.abs(W).max() > 0
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        assert np.array_equal(W, run.weights), np.abs(W - run.weights).max()
        for i in (0, N // 2, N - 1):
            assert np.array_equal(c.get_traces(i), run.sparse_trace(i)), i
        assert st["episodes"] + st2["episodes"] == ost["episodes"] and st["env_steps"] + st2["env_steps"] == N * K
        assert abs(st["sum_abs_td_error"] + st2["sum_abs_td_error"] - ost["sum_abs_td_error"]) <= 1e-4 * (1 + ost["sum_abs_td_error"])
        assert c.fx_saturations() == sat0
        with ra.Context(basis=ra.TILE_CODING, weight_mode=ra.W_SHARED, seed=7, alpha=alpha, n_envs=N, **kw) as g:      # no statistics: the captured step graphs
            g.reset()
            g.train(K - 7, want_stats=False)
            g.train(7, want_stats=False)
            assert np.array_equal(g.get_weights(), W) and np.array_equal(g.states, c.states) and np.array_equal(g.actions, c.actions)
            assert np.array_equal(g.get_traces(N - 1), c.get_traces(N - 1))
        # the reference's precision: the same rule in f64 (trajectories part ways where an argmax is decided by an fp32 rounding)
        if K <= 200:
            same = np.all(np.abs(c.states.T - r64.state) <= 1e-4 * (1 + np.abs(r64.state)), axis=1) & (c.actions == r64.action)
            assert same.mean() >= 0.85, same.mean()
            assert np.max(np.abs(W - r64.weights)) <= 0.02 * np.abs(r64.weights).max() + 1e-9
        else:                                                       # a long run on one shared table: every learner feels the first flipped argmax; as a population
            assert abs(ost["sum_abs_td_error"] - st64["sum_abs_td_error"]) <= 0.05 * st64["sum_abs_td_error"]
            assert np.max(np.abs(W - r64.weights)) <= 0.25 * np.abs(r64.weights).max()
        with pytest.raises(ra.RsrlHipError):
            c.set_traces(np.zeros((c.F, c.A), np.float32), 0)
    if "evicting" in name:                                          # the cap did take effect: some learner's list is full
        assert max(int((run.sparse_trace(i) != 0).sum()) for i in range(N)) == 512


@pytest.mark.parametrize("algo,domain,T", [(3, 1, 8), (4, 1, 8), (4, 2, 4), (3, 0, 16)])
def test_handle_is_the_driver_loops_step_on_the_callers_transitions(ra, algo, domain, T):
    # Handler::handle (sarsa_lambda.rs:63-98, q_lambda.rs:56-99) with transition i taken as learner i's: on the transitions the driver loop would have taken,
    # from the same table and the same traces, it leaves the same table and the same traces, bit for bit (and the driver loop is bitwise against the oracle)
    N = 200
    kw = dict(domain=domain, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=6, algo=algo, policy=1, epsilon=0.3, gamma=0.97, lam=0.85, trace=algo - 3,
              max_episode_steps=1000, weight_mode=ra.W_SHARED, seed=5, alpha=0.1 / T / N, n_envs=N)
    with ra.Context(**kw) as a, ra.Context(**kw) as b:
        a.reset(); b.reset()
        a.train(25); b.train(25)
        for _ in range(6):
            acts = b.actions.copy()
            frm, nxt, rew, term = b.domain_step(acts)              # b's environments advance; its agent is taught by hand
            td = b.handle(frm, acts, rew, nxt, term)
            sa = a.train(1)
            assert abs(np.abs(td).sum() - sa["sum_abs_td_error"]) <= 1e-5 * (1 + sa["sum_abs_td_error"])
            assert np.array_equal(a.get_weights(), b.get_weights())
            for i in (0, 77, N - 1):
                assert np.array_equal(a.get_traces(i), b.get_traces(i)), i
            b.states, b.actions = a.states, a.actions              # (the loop's own draws for the next action: keep the two runs on one trajectory)
            b.episode_steps = a.episode_steps
        # a batch of fewer transitions than learners: learners 0 .. M-1 learn, the others' traces stay
        z_last = b.get_traces(N - 1)
        s = b.states
        b.handle(s[:, :50].copy(), b.actions[:50].copy(), np.ones(50, np.float32), s[:, :50].copy(), np.zeros(50, np.uint8))
        assert np.array_equal(b.get_traces(N - 1), z_last) and not np.array_equal(b.get_weights(), a.get_weights())


@pytest.mark.parametrize("algo,domain,trace", [(3, 1, 0), (3, 0, 2), (4, 1, 1)])
def test_teacher_forced_vs_f64(ra, orc, algo, domain, trace):
    # the reference's precision: the f64 oracle runs the sparse-trace loop as a teacher (successor states rounded to fp32, orc_run_teacher_sparse_lambda) and every
    # batch-step's transitions go through rsrl_hip_handle (transition i = learner i): device and oracle learn from IDENTICAL inputs.  SURVEY 8(d): teacher-forced
    # max|dW| <= 1e-3 max(1, max|W|); asserted far tighter, relative to max|W| itself.  (An agent's own draw / Watkins's cut is a discrete decision a rounding can tip:
    # the seeds below have none in these horizons.)
    N, K = 192, 250
    kw = dict(domain=domain, n_tilings=8, tiles_per_dim=8, algo=algo, policy=1, epsilon=0.15, gamma=0.98, lam=0.85, trace=trace, max_episode_steps=80)
    alpha = 0.1 / 8 / N
    ag = orc.make_agent(**dict(kw, basis=orc.TILE, shared_w=True, seed=3, alpha=alpha))
    run = orc.Run(ag, N, "f64")
    run.reset()
    with ra.Context(basis=ra.TILE_CODING, weight_mode=ra.W_SHARED, seed=3, alpha=alpha, n_envs=N, **kw) as c:
        c.reset()
        td_err = 0.0
        for _ in range(K):
            t = run.teacher_step_sparse_lambda()
            frm, to = np.ascontiguousarray(t["frm"].T, dtype=np.float32), np.ascontiguousarray(t["to"].T, dtype=np.float32)
            td = c.handle(frm, t["action"], t["reward"].astype(np.float32), to, t["terminal"])
            td_err = max(td_err, float(np.max(np.abs(td - t["td"]) / (1 + np.abs(t["td"])))))
        W64 = np.array(run.weights).reshape(c.F, c.A)
        W = c.get_weights().astype(np.float64)
        wmax = np.abs(W64).max()
        assert wmax > 1e-4
        z_err = max(float(np.abs(c.get_traces(i).astype(np.float64) - run.sparse_trace(i)).max()) for i in (0, N // 3, N - 1))
        print(f"sparse lambda teacher-forced: td {td_err:.2e}  w {np.abs(W - W64).max() / wmax:.2e} of max|W| {wmax:.2e}  z {z_err:.2e}")
        assert td_err <= 2e-5, td_err                                                   # measured 2.5e-8 / 6.4e-6 (MountainCar: |W| 3.5) / 2.8e-8
        assert np.abs(W - W64).max() <= 4e-6 * wmax, (np.abs(W - W64).max(), wmax)      # measured 3.8e-7 / 1.0e-6 / 1.7e-7 of max|W|
        assert z_err <= 1e-6, z_err                                                     # measured 2.6e-7 / 2.4e-7 / 2.2e-8


def test_sparse_traces_travel_with_the_checkpoint(ra, tmp_path):
    # file version 6 / aux kind 4: whose lists (n_envs, env_offset), then every learner's sub-lists in tiling and slot order -- with lists that are FULL (the
    # slot order decides which entry the next new key overwrites), so the resumed run is the straight one bit for bit
    kw = dict(domain=0, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=3, policy=1, epsilon=0.3, gamma=0.99, lam=0.97, trace=2, max_episode_steps=0,
              weight_mode=ra.W_SHARED, seed=7, alpha=0.1 / 8 / 64, n_envs=64)
    path = str(tmp_path / "sparse.rsrlw")
    with ra.Context(**kw) as a, ra.Context(**kw) as b:
        a.reset()
        a.train(700)
        assert max(int((a.get_traces(i) != 0).sum()) for i in range(64)) == 512          # full lists at the checkpoint
        a.save_weights(path)
        s, act = a.states.copy(), a.actions.copy()
        a.train(200)
        b.reset()
        b.train(13)                                                                      # (other lists, to be replaced)
        b.load_weights(path)
        b.states, b.actions = s, act
        b.train(200)
        assert b.step_count == a.step_count == 900
        assert np.array_equal(a.get_weights(), b.get_weights())
        assert np.array_equal(a.states, b.states) and np.array_equal(a.actions, b.actions)
        for i in (0, 31, 63):
            assert np.array_equal(a.get_traces(i), b.get_traces(i)), i
        # a damaged file leaves the ctx as it was: one byte short; a length beyond the cap
        raw = open(path, "rb").read()
        w_before, z_before = b.get_weights(), b.get_traces(5)
        cut = str(tmp_path / "cut.rsrlw")
        open(cut, "wb").write(raw[:-1])
        with pytest.raises(ra.RsrlHipError):
            b.load_weights(cut)
        off = 72 + b.F * b.A * 4 + 16                                                    # header, the one shared table, u64 n_envs, u64 env_offset, then u32 len[N]
        bad = bytearray(raw); bad[off:off + 4] = (513).to_bytes(4, "little")
        open(cut, "wb").write(bytes(bad))
        with pytest.raises(ra.RsrlHipError):
            b.load_weights(cut)
        key0 = off + 4 * 64                                                              # learner 0's first key
        bad = bytearray(raw); bad[key0:key0 + 4] = (b.F * b.A).to_bytes(4, "little")
        open(cut, "wb").write(bytes(bad))
        with pytest.raises(ra.RsrlHipError):
            b.load_weights(cut)
        assert np.array_equal(b.get_weights(), w_before) and np.array_equal(b.get_traces(5), z_before)
    with ra.Context(**dict(kw, lam=0.5, n_envs=32)) as other:                            # another learner count: refused
        with pytest.raises(ra.RsrlHipError):
            other.load_weights(path)
    with ra.Context(**dict(kw, env_offset=64)) as shard:                                 # the same count, ANOTHER shard of the learners: refused, and says so
        with pytest.raises(ra.RsrlHipError, match="env_offset"):
            shard.load_weights(path)


@pytest.mark.parametrize("T,domain", [(4, 1), (16, 0)])
def test_checkpoint_file_holds_full_keys_whatever_the_lists_hold_in_memory(ra, tmp_path, T, domain):
    # in memory a key is 16 bit and relative to its tiling's slice; the FILE holds tile index * A + action (include/rsrl_hip.h): learner 0's list read back from
    # the file is the dense matrix get_traces shows, and a second ctx resumes from it bit for bit -- at 4 and at 16 tilings
    kw = dict(domain=domain, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=6, algo=3, policy=1, epsilon=0.2, gamma=0.99, lam=0.9, trace=0, max_episode_steps=0,
              weight_mode=ra.W_SHARED, seed=11, alpha=0.1 / T / 40, n_envs=40)
    path = str(tmp_path / "keys.rsrlw")
    with ra.Context(**kw) as a, ra.Context(**kw) as b:
        a.reset()
        a.train(120)
        a.save_weights(path)
        raw = open(path, "rb").read()
        off = 72 + a.F * a.A * 4 + 16
        lens = np.frombuffer(raw, np.uint32, 40, off)
        k0 = off + 4 * 40
        keys = np.frombuffer(raw, np.uint32, int(lens[0]), k0)
        vals = np.frombuffer(raw, np.float32, int(lens[0]), k0 + 4 * int(lens[0]))
        assert lens[0] > T and len(set(keys.tolist())) == len(keys) and keys.max() < a.F * a.A
        assert len(set((keys // (a.F * a.A // T)).tolist())) == T                        # entries of every tiling, as full keys
        dense = np.zeros(a.F * a.A, np.float32)
        dense[keys] = vals
        assert np.array_equal(dense.reshape(a.F, a.A), a.get_traces(0))
        s, act = a.states.copy(), a.actions.copy()
        a.train(60)
        b.reset()
        b.load_weights(path)
        b.states, b.actions = s, act
        b.train(60)
        assert np.array_equal(a.get_weights(), b.get_weights()) and np.array_equal(a.states, b.states)
        for i in (0, 17, 39):
            assert np.array_equal(a.get_traces(i), b.get_traces(i)), i


def test_sparse_lambda_is_refused_where_it_does_not_exist(ra):
    with pytest.raises(ra.RsrlHipError):                            # a shared DENSE basis has no sparse gradient
        ra.Context(domain=0, order=3, algo=3, policy=1, weight_mode=ra.W_SHARED, n_envs=8)
    with pytest.raises(ra.RsrlHipError, match="65 536"):            # a tiling's slice beyond the 16-bit slice-relative keys between the step and the trace kernel
        ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=4, tiles_per_dim=14, algo=3, policy=1, weight_mode=ra.W_SHARED, n_envs=8, alpha=0.001, lam=0.5)
    with ra.Context(domain=1, basis=ra.TILE_CODING, algo=3, policy=1, weight_mode=ra.W_SHARED, n_envs=8, alpha=0.001, lam=0.5) as c:
        c.reset()
        c.train(3)
        assert c.policy_mode(c.states).shape == (8,) and c.q_evaluate(c.states).shape == (2, 8)
