"""GPU tests of the boundary items of round 3 (all through the C ABI):
  * a caller-supplied stream is never coalesced (train() has enqueued everything when it returns);
  * rsrl_hip_group_create: all ranks in one process, a single host thread (SURVEY 8b last row; rsrl/src/core.rs:13-15);
    a sharded group reproduces the unsharded run BIT FOR BIT (the ranks exchange exact 64-bit sums);
  * exchange tags follow the number of exchanges, not the batch-step counter: a restored checkpoint that sets the counter
    back neither times out nor reads stale slots;
  * the rest of the Enumerable / Policy / Trajectory surface: find_min, expected_value (core.rs:86-116), Function<(S, A)> of
    the policies (greedy.rs:46-60, epsilon_greedy.rs:49-63, softmax.rs:84-92, random.rs:28-32), Trajectory (lib.rs:334-409);
  * QSigma's n-step backups travel with the checkpoint (resume bit-identical for n_steps > 1)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C4 = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, weight_mode=1, seed=0, max_episode_steps=200)


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def _run(c, steps=(40, 70)):
    c.reset()
    for k in steps:
        c.train(k, want_stats=False)
    c.sync()
    return c.get_weights(), c.states, c.actions


def test_caller_stream_is_never_coalesced(ra):
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    try:
        kw = dict(n_envs=4096, policy=ra.EPSILON_GREEDY, seed=3, max_episode_steps=50)
        with ra.Context(stream=stream.value, **kw) as c, ra.Context(**kw) as own:
            for x in (c, own):
                x.reset()
            for _ in range(6):                           # back-to-back short calls: what a ctx-owned stream coalesces
                c.train(20, want_stats=False)
                assert c.pending_steps == 0              # everything accepted is already on the caller's stream
                own.train(20, want_stats=False)
            assert hip.hipStreamSynchronize(stream) == 0  # the caller orders its own work with its own stream ...
            assert c.pending_steps == 0 and c.step_count == 120
            # ... and the two ctxs did the same thing, bit for bit
            assert np.array_equal(c.states, own.states) and np.array_equal(c.actions, own.actions)
            assert c.checksum() == own.checksum()
    finally:
        hip.hipStreamDestroy(stream)


@pytest.mark.parametrize("exchange", ["rccl", "peer"])
def test_group_create_of_one_is_the_plain_run(ra, exchange):
    N = 8192
    kw = dict(C4, n_envs=N, lr=0.001 / N)
    ex = ra.EXCHANGE_PEER if exchange == "peer" else ra.EXCHANGE_RCCL
    with ra.Context(**kw) as plain, ra.Context(exchange=ex, **kw) as grp:
        assert grp.comm_info() == (1, 0, -1)
        ra.Context.group_create([grp])
        assert grp.comm_info() == (1, 0, ex)
        with pytest.raises(ra.RsrlHipError):
            ra.Context.group_create([grp])               # already attached
        ref, got = _run(plain), _run(grp)
        assert np.abs(ref[0]).max() > 0
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))


def test_group_of_two_on_one_device_single_host_thread_is_the_unsharded_run_bitwise(ra):
    # two ranks = two ctxs of THIS process on the one device, driven by this one thread: each train() only enqueues its rank's
    # persistent kernel; the kernels exchange through same-process peer buffers.  Shards are whole 512-learner blocks, the
    # ranks add exact 64-bit sums => the group's weights equal the unsharded run's, bit for bit.
    N, G = 4096, 2
    kw = dict(C4, lr=0.001 / N, exchange=ra.EXCHANGE_PEER)
    ctxs = [ra.Context(n_envs=N // G, env_offset=r * (N // G), **kw) for r in range(G)]
    try:
        ra.Context.group_create(ctxs)
        assert [c.comm_info() for c in ctxs] == [(G, r, ra.EXCHANGE_PEER) for r in range(G)]
        for c in ctxs:
            c.reset()
        for k in (30, 45):
            for c in ctxs:
                c.train(k, want_stats=False)
        for c in ctxs:
            c.sync()
        W = [c.get_weights() for c in ctxs]
        with ra.Context(n_envs=N, **kw) as full:
            ref = _run(full, (30, 45))
        assert np.abs(ref[0]).max() > 0
        assert np.array_equal(W[0], W[1]) and np.array_equal(W[0], ref[0])
        assert np.array_equal(np.concatenate([c.states for c in ctxs], axis=1), ref[1])
        assert np.array_equal(np.concatenate([c.actions for c in ctxs]), ref[2])
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("persist", [True, False])
def test_exchange_survives_a_checkpoint_that_sets_the_counter_back(ra, tmp_path, persist):
    # train 12, save, train 5 more, load (the batch-step counter goes back by 5), train 9: same as a plain ctx doing the same;
    # the receive slots still hold the tags of the abandoned steps -- tags keyed by the batch-step would match them
    if not persist:
        os.environ["RSRL_NO_PERSIST"] = "1"
    try:
        N = 4096
        kw = dict(C4, n_envs=N, lr=0.001 / N)
        path = str(tmp_path / "w.bin")
        out = []
        for ex in (None, ra.EXCHANGE_PEER):
            with ra.Context(**(kw if ex is None else dict(kw, exchange=ex))) as c:
                if ex is not None:
                    ra.Context.group_create([c])
                c.reset()
                c.train(12, want_stats=False)
                c.save_weights(path)
                s, a = c.states, c.actions
                c.train(5, want_stats=False)
                c.load_weights(path)
                assert c.step_count == 12
                c.states, c.actions = s, a
                c.train(9, want_stats=False)
                c.sync()                                  # a timed-out exchange would raise here
                out.append((c.get_weights(), c.states, c.actions))
        assert np.abs(out[0][0]).max() > 0 and np.isfinite(out[1][0]).all()
        assert all(np.array_equal(a, b) for a, b in zip(*out))
    finally:
        os.environ.pop("RSRL_NO_PERSIST", None)


@pytest.mark.parametrize("cfg", [dict(domain=0, order=5, policy=1, epsilon=0.25), dict(domain=1, order=1, policy=0),
                                 dict(domain=2, order=7, policy=2, tau=0.7), dict(domain=1, basis=1, policy=3)])
def test_find_min_expected_value_policy_prob(ra, cfg):
    rng = np.random.default_rng(5)
    M = 300
    with ra.Context(n_envs=M, algo=ra.SARSA, seed=2, **cfg) as c:
        lo, hi = c.state_bounds()
        S = (lo[:, None] + (hi - lo)[:, None] * rng.random((c.D, M))).astype(np.float32)
        for i in range(0, M, 37):
            c.set_weights(rng.normal(size=(c.F, c.A)).astype(np.float32) * 0.1, i)
        c.set_weights(np.zeros((c.F, c.A), np.float32), 5)                  # ties: all-equal action values
        q = c.q_evaluate(S)
        idx, val = c.q_find_min(S)
        want = np.array([max(j for j in range(c.A) if q[j, m] == q[:, m].min()) for m in range(M)])     # ties -> LAST index
        assert np.array_equal(idx, want) and np.array_equal(val, q.min(axis=0))
        imax, _ = c.q_find_max(S)
        assert np.all(q[idx, np.arange(M)] <= q[imax, np.arange(M)])
        p = rng.random((c.A, M)).astype(np.float32)
        ev = c.q_expected_value(S, p)
        acc = np.zeros(M, np.float32)
        for b in range(c.A):                                             # fold(0.0, |acc, (x, p)| acc + x * p): fp32, not fused
            acc = (acc + (q[b] * p[b]).astype(np.float32)).astype(np.float32)
        assert np.array_equal(ev, acc)
        probs = c.policy_probs(S)
        for b in range(c.A):
            pa = c.policy_prob(S, np.full(M, b, np.int32))
            if cfg["policy"] == ra.SOFTMAX:
                assert np.array_equal(pa, q[b])                          # softmax.rs:84-92 returns the raw action value
            elif cfg["policy"] == ra.EPSILON_GREEDY:
                # epsilon_greedy.rs:49-63 `pr + (1 - eps) * p` against the vector form's `pr + p * (1 - eps)`: the same product
                assert np.array_equal(pa, probs[b])
            else:
                assert np.array_equal(pa, probs[b])
        with pytest.raises(ra.RsrlHipError):
            c.policy_prob(S, np.full(M, c.A, np.int32))                  # a host action outside [0, A) is refused


@pytest.mark.parametrize("cfg", [dict(domain=0, order=5), dict(domain=1, order=7), dict(domain=1, basis=1)])
def test_rollout_trajectory_is_the_rollout(ra, cfg):
    M, L = 64, 120
    with ra.Context(n_envs=M, policy=ra.EPSILON_GREEDY, epsilon=0.2, seed=4, max_episode_steps=80, **cfg) as c:
        c.reset()
        c.train(300, want_stats=False)
        n_states, tot = c.rollout_greedy(L)
        tr = c.rollout_trajectory(L)
        assert np.array_equal(tr["n_states"], n_states) and np.array_equal(tr["total_reward"], tot)
        with ra.Context(n_envs=M, **cfg) as env:                          # replay the recorded actions through Domain::transition
            env.domain_reset()
            assert np.array_equal(tr["states"][0], env.states)           # Trajectory.start = Domain::default()
            alive = np.ones(M, bool)
            for k in range(L - 1):
                alive &= (k + 1) < n_states
                frm, nxt, rew, term = env.domain_step(tr["actions"][k])
                assert np.array_equal(nxt[:, alive], tr["states"][k + 1][:, alive])
                assert np.array_equal(rew[alive], tr["rewards"][k][alive])
                last = alive & (n_states == k + 2)
                assert np.array_equal(term[last] != 0, tr["terminal"][last] != 0)
                assert not term[alive & ~last].any()                     # only the last observation can be terminal
                # rows past the end stay zero
                assert not tr["states"][k + 1][:, ~alive].any() and not tr["rewards"][k][~alive].any()
        # the actions are policy.mode of the recorded states
        for m in range(0, M, 17):
            n = int(n_states[m]) - 1
            if n > 0:
                Sm = np.ascontiguousarray(tr["states"][:n, :, m].T)
                # policy_mode evaluates state j with learner j's weights: give every learner learner m's states one at a time
                W = c.get_weights(m)
                with ra.Context(n_envs=n, policy=ra.EPSILON_GREEDY, **cfg) as probe:
                    probe.set_weights_all(W)
                    assert np.array_equal(probe.policy_mode(Sm), tr["actions"][:n, m])
        sub = c.rollout_trajectory(L, M=10)
        assert np.array_equal(sub["n_states"], n_states[:10]) and np.array_equal(sub["states"], tr["states"][:, :, :10])
        one = c.rollout_trajectory(1)                                     # Some(1): take(0) -- no transition is kept
        assert np.all(one["n_states"] == 1) and np.all(one["total_reward"] == 0) and not one["terminal"].any()


def test_qsigma_backups_travel_with_the_checkpoint(ra, tmp_path):
    # (no step cap: the episode counters are not part of a checkpoint)
    kw = dict(n_envs=512, algo=ra.Q_SIGMA, policy=ra.EPSILON_GREEDY, epsilon=0.1, sigma=0.5, n_steps=3, alpha=0.1, lr=1.0, seed=9)
    path = str(tmp_path / "qs.bin")
    with ra.Context(**kw) as a:
        a.reset()
        a.train(130, want_stats=False)
        a.save_weights(path)
        s, act = a.states, a.actions
        a.train(70, want_stats=False)
        ref = (a.checksum()[0], a.states, a.get_weights(17))
    with ra.Context(**kw) as b:
        b.load_weights(path)
        b.states, b.actions = s, act
        b.train(70, want_stats=False)
        got = (b.checksum()[0], b.states, b.get_weights(17))
    assert ref[0] == got[0] and np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2])
    with ra.Context(**dict(kw, n_steps=2)) as other:
        with pytest.raises(ra.RsrlHipError):
            other.load_weights(path)                                     # another ring geometry: the size check refuses it


def test_fixed_point_saturation_is_counted(ra):
    # a healthy shared-W run clamps nothing; a diverged one (huge step size: |lr*e*phi| beyond 2^14 * 2^floor(log2 lr)) is counted
    kw = dict(C4, n_envs=1024)
    with ra.Context(lr=1e-6, **kw) as ok:
        base = ok.fx_saturations()
        ok.reset(); ok.train(50, want_stats=False); ok.sync()
        assert ok.fx_saturations() == base
    with ra.Context(lr=1e-30, **kw) as bad:                        # lsb = 2^-98 (the floor): any |term| > 2^-56 clamps
        bad.reset(); bad.train(5, want_stats=False); bad.sync()
        bad.set_weights(np.full((bad.F, bad.A), 1e20, np.float32))
        bad.train(5, want_stats=False); bad.sync()
        assert bad.fx_saturations() > base


@pytest.mark.parametrize("cfg", [dict(domain=0, order=2), dict(domain=0, order=4), dict(domain=1, order=1), dict(domain=2, order=1, algo=1),
                                 dict(domain=0, order=5, algo=2, policy=2, tau=0.8), dict(domain=0, order=3, algo=5, alpha=0.5)])
@pytest.mark.parametrize("N", [700, 3000])
def test_persistent_kernel_equals_one_launch_per_step_on_every_shape(ra, cfg, N):
    # the persistent shared-W kernel against the one-launch-per-batch-step path (itself bitwise against the oracle): odd A*F
    # (order 2: 27 entries -> a padding granule; order 4: 75), A = 2 (CartPole), a ragged last block, fewer blocks than
    # granule pairs (owners loop), every one-step agent family; launches of 1 / 9 / 33 batch-steps
    kw = dict(dict(algo=0, policy=1, epsilon=0.15, gamma=0.95, weight_mode=1, seed=5, max_episode_steps=30, n_envs=N, lr=0.02 / N), **cfg)

    def run():
        with ra.Context(**kw) as c:
            c.reset()
            st = [c.train(k) for k in (1, 9, 33)]
            c.sync()
            return c.get_weights(), c.states, c.actions, [s["episodes"] for s in st], [s["sum_episode_steps"] for s in st]
    got = run()
    os.environ["RSRL_NO_PERSIST"] = "1"
    try:
        ref = run()
    finally:
        os.environ.pop("RSRL_NO_PERSIST", None)
    assert np.abs(ref[0]).max() > 0
    assert all(np.array_equal(a, b) for a, b in zip(ref[:3], got[:3]))
    assert ref[3] == got[3] and ref[4] == got[4]


def test_empty_oversized_and_ragged_batches(ra):
    # the granular calls take 1 <= M <= n_envs items; an empty or an oversized batch is refused with EINVAL and changes nothing,
    # a ragged one (M < n_envs) touches learners 0..M-1 only; train(0) is a no-op
    N = 200
    with ra.Context(n_envs=N, policy=1, epsilon=0.1, seed=2) as c:
        c.reset(); c.train(10)
        before = c.checksum(); s0 = c.states.copy()
        L, h = c._L, c._h
        buf = np.zeros((2, N + 8), np.float32); out = np.zeros((3, N + 8), np.float32)
        idx = np.zeros(N + 8, np.int32)
        p = lambda a: a.ctypes.data_as(__import__("ctypes").c_void_p)       # noqa: E731
        for M in (0, -3, N + 1):
            assert L.rsrl_hip_q_evaluate(h, p(buf), M, p(out)) != 0
            assert L.rsrl_hip_policy_sample(h, p(buf), M, p(idx)) != 0
            assert L.rsrl_hip_handle(h, p(buf), p(idx), p(out), p(buf), p(idx), M, p(out)) != 0
        assert c.checksum() == before and np.array_equal(c.states, s0)
        st = c.train(0)
        assert st["env_steps"] == 0 and c.checksum() == before
        # ragged handle: 7 transitions move 7 learners
        a = c.actions[:7].copy()
        frm = c.states[:, :7].copy()
        w8 = c.get_weights(8).copy(); w3 = c.get_weights(3).copy()
        td = c.handle(frm, a, np.full(7, -1.0, np.float32), frm, np.zeros(7, np.uint8))
        assert td.shape == (7,) and np.array_equal(c.get_weights(8), w8) and not np.array_equal(c.get_weights(3), w3)
        with pytest.raises(ra.RsrlHipError):
            c.train(-1)
