"""GPU tests of round 4's boundary additions, all through the C ABI:
  * the per-learner epsilon schedule (`agent.policy.epsilon *= 0.995` once per EPISODE of the learner, examples/sarsa_lambda.rs:48-75,
    :68; pub field epsilon_greedy.rs:19) -- bitwise against the oracle's f32d run on every kernel family that runs it, the
    reference example's configuration over >= 50 episodes per learner included, and N = 1 against the f64 (reference) loop;
  * Domain::rollout under any of the four policies (lib.rs:448-479 takes any FnMut(&S) -> A) -- bitwise against the oracle;
  * the persistent shared-W kernel's co-residency guards: unrelated persistent ctxs of one process on one device, the collective
    decision of a peer group, rsrl_hip_group_train."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


# ---------------------------------------------------------------------------------------------------------------------------
# per-learner epsilon schedule
# ---------------------------------------------------------------------------------------------------------------------------
SCHED = [
    # rsrl/examples/sarsa_lambda.rs itself: SARSALambda, Fourier(5), replacing traces, alpha 0.01, gamma 0.99, lambda 0.7, eps 0.2 x 0.995
    ("sarsa(lambda), the reference example", dict(domain=0, order=5, algo=3, policy=1, trace=1, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2), 0.995, 0.0, "hook"),
    ("q(lambda), floor", dict(domain=0, order=5, algo=4, policy=1, trace=0, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.3), 0.9, 0.05, "hook"),
    ("q-learning (k_train_reg)", dict(domain=0, order=5, algo=0, policy=1, gamma=0.9, lr=0.001, epsilon=0.3), 0.97, 0.0, "dev"),
    ("sarsa (agent shares the policy object)", dict(domain=0, order=3, algo=1, policy=1, gamma=0.9, lr=0.002, epsilon=0.4), 0.95, 0.02, "dev"),
    ("expected sarsa, CartPole", dict(domain=1, order=1, algo=2, policy=1, gamma=0.95, lr=0.001, alpha=0.5, epsilon=0.3), 0.98, 0.0, "dev"),
    ("sarsa, own greedy agent policy", dict(domain=0, order=5, algo=1, policy=1, gamma=0.9, lr=0.001, epsilon=0.5, agent_policy=0), 0.96, 0.0, "dev"),
    ("sarsa on tile coding (k_train_mem)", dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=1, policy=1, gamma=0.99, lr=0.0125, epsilon=0.3), 0.9, 0.0, "hook"),
]


@pytest.mark.parametrize("name,kw,decay,floor,loop", SCHED, ids=[c[0] for c in SCHED])
def test_epsilon_schedule_bitwise(ra, orc, name, kw, decay, floor, loop):
    N, K, cap = 96, 900, 15                                     # 900 / 15 = 60 episodes per learner at least
    okw = dict(kw)
    ag = orc.make_agent(seed=5, max_episode_steps=cap, epsilon_decay=decay, epsilon_min=floor, **okw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    ost = (run.train if loop == "hook" else run.train_dev)(K)
    with ra.Context(n_envs=N, seed=5, max_episode_steps=cap, epsilon_decay=decay, epsilon_min=floor, **kw) as c:
        c.reset()
        st = [c.train(k) for k in (1, 299, 600)]                # any split into launches
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        assert np.array_equal(c.epsilons, run.eps.astype(np.float32))
        for i in (0, 1, 47, 95):
            assert np.array_equal(c.get_weights(i), run.weights[i]), i
            if kw["algo"] in (3, 4):
                assert np.array_equal(c.get_traces(i), run.traces[i]), i
        assert sum(s["episodes"] for s in st) == ost["episodes"] >= 60 * N
    assert run.eps.max() < np.float32(kw["epsilon"]) and run.eps.min() >= np.float32(floor)       # it did decay, never below the floor
    if floor > 0:
        assert (run.eps == np.float32(floor)).any()


def test_epsilon_schedule_one_learner_follows_the_reference_loop(ra, orc):
    # N = 1 against the f64 oracle (the reference's arithmetic): the device's fp32 epsilon stays within rounding of the f64
    # schedule eps0 * 0.995^k, k = episodes finished, and epsilon changes exactly when an episode ends
    kw = dict(domain=0, order=5, algo=3, policy=1, trace=1, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)
    with ra.Context(n_envs=1, seed=0, max_episode_steps=25, epsilon_decay=0.995, **kw) as c:
        c.reset()
        eps, episodes, prev = 0.2, 0, np.float32(0.2)
        for _ in range(120):
            st = c.train(5)
            now = c.epsilons[0]
            if st["episodes"]:
                for _k in range(st["episodes"]):
                    eps *= 0.995                                 # examples/sarsa_lambda.rs:68, f64
                episodes += st["episodes"]
                assert now < prev
            else:
                assert now == prev
            assert abs(float(now) - eps) <= (episodes + 2) * 6e-8 * 0.2 + 1e-9          # one fp32 rounding per episode
            prev = now
        assert episodes >= 20


def test_epsilon_schedule_api(ra, tmp_path):
    kw = dict(domain=0, order=5, algo=3, policy=1, trace=1, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2, n_envs=64, seed=2,
              max_episode_steps=20)
    with ra.Context(epsilon_decay=0.99, **kw) as c, ra.Context(epsilon_decay=0.99, **kw) as d, ra.Context(**kw) as plain:
        assert np.all(plain.epsilons == np.float32(0.2))        # no schedule: N copies of the ctx's value
        for x in (c, d):
            x.reset()
        c.train(100)
        e = c.epsilons
        assert e.max() < 0.2 and e.min() >= np.float32(0.2) * np.float32(0.99) ** 6
        # the schedule's state travels in the checkpoint (file version 4): a resumed run continues it bit for bit
        path = str(tmp_path / "eps.ckpt")
        c.save_weights(path)
        import struct
        raw = open(path, "rb").read()
        assert struct.unpack_from("<I", raw, 8)[0] == 4 and len(raw) == 72 + 2 * 64 * 36 * 3 * 4 + 64 * 4
        assert np.array_equal(np.frombuffer(raw, dtype="<f4", count=64, offset=len(raw) - 256), e)
        c.reset(); c.train(60)
        d.load_weights(path)
        assert np.array_equal(d.epsilons, e) and d.step_count == 100
        d.reset(); d.train(60)
        assert np.array_equal(d.states, c.states) and np.array_equal(d.epsilons, c.epsilons) and (c.epsilons < e).any()
        for i in (0, 63):
            assert np.array_equal(d.get_weights(i), c.get_weights(i)) and np.array_equal(d.get_traces(i), c.get_traces(i))
        with pytest.raises(ra.RsrlHipError):                     # a file with the schedule's section does not fit a ctx without one
            plain.load_weights(path)
        c.set_epsilon(0.5)                                       # the pub field of every learner
        assert np.all(c.epsilons == np.float32(0.5))
        # policy queries address learner m's own policy object
        s = c.states
        c.train(40)
        e = c.epsilons
        p = c.policy_probs(s)
        assert np.allclose(p.min(axis=0), e / 3, rtol=1e-6)
    for bad in (dict(epsilon_decay=0.0), dict(epsilon_decay=1.5), dict(epsilon_decay=0.9, epsilon_min=-1.0),
                dict(epsilon_decay=0.9, policy=0), dict(epsilon_decay=0.9, weight_mode=1), dict(epsilon_decay=0.9, steps_per_launch=1, algo=0),
                dict(epsilon_decay=0.9, algo=6, lr_td=0.01), dict(epsilon_decay=0.9, algo=6, lr_td=0.01, domain=2, order=7)):      # (round 6: the one-step and lambda agents run it on the wave family)
        with pytest.raises(ra.RsrlHipError):
            ra.Context(**{**dict(n_envs=8, policy=1), **bad})


# ---------------------------------------------------------------------------------------------------------------------------
# Domain::rollout under any policy
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("policy,eps,tau", [(1, 0.3, 1.0), (2, 0.0, 0.7), (3, 0.0, 1.0), (0, 0.0, 1.0)], ids=["egreedy", "softmax", "random", "greedy"])
@pytest.mark.parametrize("family", ["fourier", "tiles"])
def test_rollout_policy_bitwise(ra, orc, policy, eps, tau, family):
    N, L = 160, 120
    kw = dict(domain=0, order=5, algo=0, policy=1, gamma=0.9, lr=0.002, epsilon=0.2) if family == "fourier" else \
        dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=1, policy=1, gamma=0.99, lr=0.0125, epsilon=0.2)
    ag = orc.make_agent(seed=3, max_episode_steps=60, **kw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    (run.train_dev if family == "fourier" else run.train)(300)
    with ra.Context(n_envs=N, seed=3, max_episode_steps=60, **kw) as c:
        c.reset()
        c.train(300)
        for i in (0, N - 1):
            assert np.array_equal(c.get_weights(i), run.weights[i])
        before = (c.states, c.actions)
        for call in range(2):                                    # successive calls: independent draws, each reproducible
            tr = c.rollout_policy(policy, L, epsilon=eps, tau=tau)
            n_o, tot_o, act_o = run.rollout_policy(policy, L, epsilon=eps, tau=tau, call=call)
            assert np.array_equal(tr["n_states"], n_o)
            assert np.array_equal(tr["total_reward"], tot_o)
            for i in range(N):
                k = int(n_o[i]) - 1
                assert np.array_equal(tr["actions"][:k, i], act_o[:k, i]), (call, i)
            if call == 0:
                first = tr
        if policy in (1, 2, 3):
            assert not np.array_equal(first["actions"], tr["actions"])       # another stream of draws
        sub = c.rollout_policy(policy, L, M=7, epsilon=eps, tau=tau)          # (call 2: its own stream)
        assert sub["n_states"].shape == (7,)
        assert np.array_equal(c.states, before[0]) and np.array_equal(c.actions, before[1])      # the training envs are untouched
        # the trajectory is a trajectory: replaying its actions through Domain::transition reproduces its states
        k = int(first["n_states"][5]) - 1
        s = first["states"][0, :, 5].astype(np.float64)
        for j in range(min(k, 20)):
            s2, r, term = orc.domain_step(kw["domain"], s.astype(np.float32), int(first["actions"][j, 5]), prec="f32d")
            assert np.array_equal(s2, first["states"][j + 1, :, 5]) and r == first["rewards"][j, 5]
            s = s2
    with ra.Context(n_envs=4) as c:
        for bad in (dict(policy=7), dict(policy=1, epsilon=1.5), dict(policy=2, tau=0.0)):
            with pytest.raises(ra.RsrlHipError):
                c.rollout_policy(bad.pop("policy"), 10, **bad)
        with pytest.raises(ra.RsrlHipError):
            c.rollout_policy(1, 0)


def test_rollout_policy_limits(ra):
    # epsilon = 0 is the greedy sample (ties aside: none with learned weights), epsilon = 1 the Random policy's rollout
    with ra.Context(n_envs=512, policy=1, epsilon=0.1, seed=1, lr=0.002, max_episode_steps=200) as c:
        c.reset()
        c.train(600)
        g = c.rollout_trajectory(150)
        e0 = c.rollout_policy(1, 150, epsilon=0.0)
        same = (g["n_states"] == e0["n_states"]).mean()
        assert same >= 0.98                                      # Greedy.sample vs Greedy.mode differ only on (near-)ties
        e1 = c.rollout_policy(1, 150, epsilon=1.0)
        a = e1["actions"][0]
        assert abs(np.bincount(a, minlength=3) / a.size - 1 / 3).max() < 0.08
        sm = c.rollout_policy(2, 150, tau=1e-3)                  # a cold softmax is (almost) greedy
        assert (sm["n_states"] == g["n_states"]).mean() >= 0.9


# ---------------------------------------------------------------------------------------------------------------------------
# the persistent shared-W kernel: co-residency by construction
# ---------------------------------------------------------------------------------------------------------------------------
C4 = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, weight_mode=1, seed=0, max_episode_steps=200)


def _run(c, steps):
    c.reset()
    for k in steps:
        c.train(k, want_stats=False)
    c.sync()
    return c.get_weights(), c.states, c.actions


def test_unrelated_persistent_ctxs_share_a_device(ra):
    # two independent shared-W ctxs, each a full one-block-per-CU grid (131 072 learners), stepped from two threads at once: without
    # the gate both grids could become partially resident and wait for their own missing blocks until the timeout.  Each must
    # finish, without an error, with exactly the weights of a run on its own.
    N = 131072
    kw = dict(C4, n_envs=N, lr=0.001 / N, peer_timeout_ms=3000)
    with ra.Context(**kw) as solo:
        ref = _run(solo, (24, 24, 24))
    ctxs = [ra.Context(**kw) for _ in range(2)]
    out, errs = [None, None], []

    def work(j):
        try:
            out[j] = _run(ctxs[j], (24, 24, 24))
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(j,)) for j in range(2)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs, errs
    for o in out:
        assert o is not None and all(np.array_equal(a, b) for a, b in zip(ref, o))
    for c in ctxs:
        c.close()


def test_group_decides_the_path_collectively(ra):
    # two PEER ranks on ONE device whose grids do not fit together (2 x 256 blocks): every rank must take the per-step path -- no
    # timeout, replicas identical, and the same mini-batch rule as the unsharded run.  One thread drives both through group_train.
    N = 131072
    kw = dict(C4, lr=0.001 / (2 * N), exchange=ra.EXCHANGE_PEER, peer_timeout_ms=3000)
    a, b = ra.Context(n_envs=N, **kw), ra.Context(n_envs=N, env_offset=N, **kw)
    ra.Context.group_create([a, b])
    for c in (a, b):
        c.reset()
    ra.Context.group_train([a, b], 40)
    ra.Context.group_train([a, b], 33)
    for c in (a, b):
        c.sync()
    assert a.timing_read()[2] != "k_shared_persist" and b.timing_read()[2] != "k_shared_persist"
    wa, wb = a.get_weights(), b.get_weights()
    assert np.array_equal(wa, wb) and np.abs(wa).max() > 0
    with ra.Context(n_envs=2 * N, **dict(kw, exchange=ra.EXCHANGE_RCCL)) as full:
        ref = _run(full, (40, 33))
    assert np.max(np.abs(ref[0] - wa)) <= 2e-6 * max(1.0, np.abs(ref[0]).max())
    a.close(); b.close()
    # small ranks fit together: the persistent kernel, and then the sharded run IS the unsharded one (exact integer totals)
    n = 2048
    kw = dict(C4, lr=0.001 / (2 * n), exchange=ra.EXCHANGE_PEER)
    a, b = ra.Context(n_envs=n, **kw), ra.Context(n_envs=n, env_offset=n, **kw)
    ra.Context.group_create([a, b])
    for c in (a, b):
        c.reset()
    ra.Context.group_train([a, b], 50)
    for c in (a, b):
        c.sync()
    assert a.timing_read()[2] == "k_shared_persist"
    with ra.Context(n_envs=2 * n, **dict(kw, exchange=ra.EXCHANGE_RCCL)) as full:
        ref = _run(full, (50,))
    assert np.array_equal(a.get_weights(), ref[0]) and np.array_equal(b.get_weights(), ref[0])
    a.close(); b.close()


def test_no_persist_on_one_rank_counts_for_the_group(ra, tmp_path):
    # RSRL_NO_PERSIST in ONE rank's environment travels in its handle: both ranks take the per-step path (with different paths they
    # would wait for each other until the timeout)
    import json
    import subprocess
    import sys
    script = tmp_path / "np.py"
    script.write_text(r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["RSRL_ROOT"])
import rsrl_amd as ra
kw = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, weight_mode=1, seed=0, max_episode_steps=200, lr=0.001 / 4096,
          exchange=ra.EXCHANGE_PEER, peer_timeout_ms=3000)
a = ra.Context(n_envs=2048, **kw)
ha = a.peer_export(2)
os.environ["RSRL_NO_PERSIST"] = "1"
b = ra.Context(n_envs=2048, env_offset=2048, **kw)
hb = b.peer_export(2)
del os.environ["RSRL_NO_PERSIST"]
a.peer_connect([ha, hb], 0); b.peer_connect([ha, hb], 1)
for c in (a, b):
    c.reset()
ra.Context.group_train([a, b], 40)
a.sync(); b.sync()
print("RESULT " + json.dumps({"same": bool(np.array_equal(a.get_weights(), b.get_weights())), "ka": a.timing_read()[2], "kb": b.timing_read()[2],
                              "absw": float(np.abs(a.get_weights()).max())}), flush=True)
os._exit(0)
''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, str(script)], env=dict(os.environ, RSRL_ROOT=root), capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "RESULT " in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert d["same"] and d["absw"] > 0 and d["ka"] != "k_shared_persist" and d["kb"] != "k_shared_persist", d


def test_single_thread_rccl_group_is_stepped_by_group_train(ra):
    # a single-thread RCCL group needs one device per rank: on this one-GPU box the group has one rank, which rsrl_hip_train may
    # still step (no second communicator to starve); group_train runs the grouped sequence and gives the same weights
    N = 8192
    kw = dict(C4, n_envs=N, lr=0.001 / N)
    with ra.Context(**kw) as plain, ra.Context(**kw) as g1:
        ra.Context.group_create([g1])
        ref = _run(plain, (30, 41))
        g1.reset()
        ra.Context.group_train([g1], 30)
        ra.Context.group_train([g1], 41)
        g1.sync()
        assert np.array_equal(g1.get_weights(), ref[0]) and np.array_equal(g1.states, ref[1])
        with pytest.raises(ra.RsrlHipError):
            ra.Context.group_train([plain], 3)                   # not a group
    if ra.device_count() >= 2:                                   # (never on the build box; the driver's multi-GPU node)
        a, b = ra.Context(**dict(kw, n_envs=N // 2, device=0)), ra.Context(**dict(kw, n_envs=N // 2, env_offset=N // 2, device=1))
        ra.Context.group_create([a, b])
        with pytest.raises(ra.RsrlHipError):
            a.train(3)                                           # un-grouped collectives from one thread: refused
        for c in (a, b):
            c.reset()
        ra.Context.group_train([a, b], 30)
        a.sync(); b.sync()
        assert np.array_equal(a.get_weights(), b.get_weights())
        a.close(); b.close()


def test_old_qsigma_checkpoint_is_still_read(ra, tmp_path):
    # files written before QSigma's backups travelled: version 2, aux_kind 0 -- the weights load, the backups start empty
    import struct
    kw = dict(domain=0, order=3, algo=9, policy=1, epsilon=0.2, gamma=0.9, lr=0.01, alpha=0.5, sigma=0.5, n_steps=3, n_envs=16, seed=1,
              max_episode_steps=30)
    with ra.Context(**kw) as c, ra.Context(**kw) as d:
        c.reset()
        c.train(50)
        new = str(tmp_path / "v3.ckpt")
        c.save_weights(new)
        raw = open(new, "rb").read()
        F, A = c.F, c.A
        body = 16 * F * A * 4
        hdr = bytearray(raw[:72])
        assert struct.unpack_from("<I", hdr, 8)[0] == 3 and struct.unpack_from("<i", hdr, 12 + 10 * 4)[0] == 3
        struct.pack_into("<I", hdr, 8, 2)                         # version 2
        struct.pack_into("<i", hdr, 12 + 10 * 4, 0)               # aux_kind 0
        old = str(tmp_path / "v2.ckpt")
        open(old, "wb").write(bytes(hdr) + raw[72:72 + body])
        d.load_weights(old)
        for i in (0, 15):
            assert np.array_equal(d.get_weights(i), c.get_weights(i))
        d.reset()
        d.train(10)                                               # runs from empty backups
        assert np.all(np.isfinite(d.get_weights(0)))


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_two_ranks_on_one_gpu_has_no_error_leg(tmp_path, ranks):
    # the N > 1 bench path end to end on the one-GPU box: two processes (torch.distributed.run), env-sharded fused loop, the streaming
    # leg, the shared-W peer exchange between PROCESSES that share the device (hipIpc; the group decides collectively which kernels fit)
    # -- no leg may carry an error, both ranks must be seen, the replicas of W must agree, and the CPU baseline is there for N > 1 too
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo")
    # ranks = 8: the world size of a node (the largest ever exercised was 4 in-process / 2 processes): eight processes share the one
    # device, the peer group of eight decides collectively which kernels fit it, hop 2 fans out to eight hipIpc-mapped receive buffers
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--allow-oversubscribe", "--envs", "4096", "--steps", "20", "--warmup", "5",
                        "--region-seconds", "0.2", "--regions", "3", "--detail", str(tmp_path / "detail.json")], env=env, capture_output=True, text=True,
                       timeout=900, cwd=str(tmp_path))
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert p.returncode == 0 and lines, (p.stdout[-1500:], p.stderr[-3000:])
    # the driver's line: the LAST stdout line, one compact JSON object (round 5's 24 KB line could not be parsed); the full record is the detail file
    line = json.loads(lines[-1])
    assert len(lines[-1]) < 8192 and line["value"] > 0 and line["n_gpus"] == 1 and "roofline" in line and "cpu_baseline" in line
    assert line["legs"]["shared_w"]["value"] > 0 and line["detail"].endswith("detail.json")
    d = json.load(open(tmp_path / "detail.json"))
    assert d["value"] == pytest.approx(line["value"], rel=1e-5)
    assert d["ranks_seen"] == ranks and d["n_gpus"] == 1 and "oversubscribed" in d and d["value"] > 0
    assert [r["rank"] for r in d["per_rank"]] == list(range(ranks)) and all(r["kernel_us_per_batch_step"] > 0 and r["device"] == 0 for r in d["per_rank"])
    assert len({r["device_identity"] for r in d["per_rank"]}) == 1

    def errors(x, path=""):
        if isinstance(x, dict):
            for k, v in x.items():
                if k == "error" and v:
                    yield path + "/error: " + str(v)
                yield from errors(v, path + "/" + k)
    if any("timed out" in e for e in errors(d)) and os.environ.get("RSRL_ALLOW_TIMESLICE_SKIP") == "1":      # opt-in only: a time-out fails by default
        pytest.skip("processes are time-sliced exclusively on this GPU: " + repr(list(errors(d))))
    assert not list(errors(d)), list(errors(d))
    assert d["shared_w"]["replicas_consistent"] and d["shared_w"]["exchange_world_size"] == ranks and d["shared_w"]["ranks"] == ranks
    assert len(d["shared_w"]["per_rank_kernel_us_per_batch_step"]) == ranks
    assert d["shared_w"]["exchange_kind"] == "peer" and "skipped" in d["shared_w_rccl"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["measured_with_ranks"] == ranks
    # the fraction is printed only while the committed profile belongs to the loaded binary (bench.profile_digest)
    if d["roofline"]["profile_digest_matches"]:
        assert 0 < d["roofline"]["frac"] <= 1 and 0 < d["roofline"]["useful_frac"] < d["roofline"]["frac"]
    else:
        assert d["roofline"]["frac"] is None and "frac" in d["roofline"]["from_stale_profile"]


def test_exchange_auto_prefers_the_peer_exchange(ra):
    # config.exchange defaults to AUTO: group_create takes the one-hop peer exchange whenever every device reaches every other's memory
    # (trivially so on one device), comm_init makes the ctx an RCCL rank, peer_export a PEER rank -- and all three are the plain run
    N = 8192
    kw = dict(C4, n_envs=N, lr=0.001 / N)
    assert ra.can_access_peer(0, 0)
    with pytest.raises(ra.RsrlHipError):
        ra.can_access_peer(0, 99)
    with ra.Context(**kw) as plain, ra.Context(**kw) as grp, ra.Context(**kw) as rccl, ra.Context(**kw) as peer:
        assert grp.cfg.exchange == ra.EXCHANGE_AUTO and grp.comm_info() == (1, 0, -1)
        ra.Context.group_create([grp])
        assert grp.comm_info() == (1, 0, ra.EXCHANGE_PEER)
        rccl.comm_init(ra.Context.comm_unique_id(), 1, 0)
        assert rccl.comm_info() == (1, 0, ra.EXCHANGE_RCCL)
        peer.peer_connect([peer.peer_export(1)], 0)
        assert peer.comm_info() == (1, 0, ra.EXCHANGE_PEER)
        ref = _run(plain, (40, 75))                              # (75: plain steps, then 30-step graph replays at t = 0 mod 3 on the RCCL path)
        for other in (grp, rccl, peer):
            got = _run(other, (40, 75))
            assert all(np.array_equal(a, b) for a, b in zip(ref, got))
        with pytest.raises(ra.RsrlHipError):
            rccl.peer_export(1)                                  # decided


# ---------------------------------------------------------------------------------------------------------------------------
# GreedyGQ and QSigma off the register family: tile coding (per-learner tables) and the generic Fourier orders.  The reference's agents
# are generic over the approximator (greedy_gq.rs:49-71, q_sigma.rs:80-105).
# ---------------------------------------------------------------------------------------------------------------------------
WIDE = [
    ("greedy_gq, CartPole tiles 8 x 8^4", dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=6, policy=1, gamma=0.99, lr=0.0125, lr_td=0.001, epsilon=0.1)),
    ("greedy_gq, MountainCar tiles 4 x 6^2", dict(domain=0, basis=1, n_tilings=4, tiles_per_dim=6, algo=6, policy=1, gamma=0.99, lr=0.025, lr_td=0.002, epsilon=0.1)),
    ("greedy_gq, MountainCar Fourier(6)", dict(domain=0, order=6, algo=6, policy=1, gamma=0.99, lr=0.05, lr_td=0.001, epsilon=0.1)),
    ("greedy_gq, CartPole Fourier(2)", dict(domain=1, order=2, algo=6, policy=2, tau=0.5, gamma=0.99, lr=0.02, lr_td=0.001)),
    ("q_sigma n=3, CartPole tiles 8 x 8^4", dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=9, policy=1, gamma=0.95, lr=0.0125, alpha=0.5, sigma=0.5, n_steps=3, epsilon=0.2)),
    ("q_sigma n=1 sigma=1, Acrobot tiles 4 x 4^4", dict(domain=2, basis=1, n_tilings=4, tiles_per_dim=4, algo=9, policy=1, gamma=0.95, lr=0.025, alpha=1.0, sigma=1.0, n_steps=1, epsilon=0.2)),
    ("q_sigma n=4 tree backup, MountainCar Fourier(7)", dict(domain=0, order=7, algo=9, policy=1, gamma=0.9, lr=0.01, alpha=0.5, sigma=0.0, n_steps=4, epsilon=0.2)),
    ("q_sigma n=2, Acrobot Fourier(2)", dict(domain=2, order=2, algo=9, policy=3, gamma=0.9, lr=0.01, alpha=0.5, sigma=0.5, n_steps=2)),
]


@pytest.mark.parametrize("name,kw", WIDE, ids=[c[0] for c in WIDE])
def test_gq_and_qsigma_off_the_register_family_bitwise(ra, orc, name, kw, tmp_path):
    N, K = 40, 260
    ag = orc.make_agent(seed=13, max_episode_steps=30, **kw)
    run = orc.Run(ag, N, "f32d")
    run.reset()
    ost = run.train(K)
    with ra.Context(n_envs=N, seed=13, max_episode_steps=30, **kw) as c:
        c.reset()
        st = [c.train(k) for k in (90, 1, K - 91)]
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        for i in (0, 1, 19, N - 1):
            assert np.array_equal(c.get_weights(i), run.weights[i]), i
            if kw["algo"] == 6:
                assert np.array_equal(c.get_td_weights(i), run.traces[i]), i
        assert sum(s["episodes"] for s in st) == ost["episodes"] > 0
        assert np.abs(run.weights).max() > 0 and np.all(np.isfinite(run.weights))
        # Handler::handle on caller-supplied transitions goes through the same code (teacher forcing): once more, bit for bit
        s, a = c.states, c.actions
        frm, nxt, rew, term = c.domain_step(a)
        term[::4] = 1
        Wb = [c.get_weights(i) for i in range(N)]
        Vb = [c.get_td_weights(i) for i in range(N)] if kw["algo"] == 6 else None
        td = c.handle(frm, a, rew, nxt, term)
        if kw["algo"] == 6:
            for i in (0, 7, N - 1):
                W, V = Wb[i].copy(), Vb[i].copy()
                d = orc.handle_gq(ag, W, V, frm[:, i], a[i], rew[i], nxt[:, i], term[i], "f32d")
                assert np.float32(d) == td[i] and np.array_equal(c.get_weights(i), W) and np.array_equal(c.get_td_weights(i), V), i
        # checkpoint: the second approximator / the n-step backups travel, a resumed run is the uninterrupted one
        path = str(tmp_path / "wide.ckpt")
        c.save_weights(path)
        c.reset(); c.train(40)
        ref = (c.states, [c.get_weights(i) for i in (0, N - 1)])
        with ra.Context(n_envs=N, seed=13, max_episode_steps=30, **kw) as d2:
            d2.load_weights(path)
            d2.reset(); d2.train(40)
            assert np.array_equal(d2.states, ref[0]) and all(np.array_equal(d2.get_weights(i), w) for i, w in zip((0, N - 1), ref[1]))
    for bad in (dict(algo=6, lr_td=0.01, basis=1, weight_mode=1), dict(algo=9, basis=1, weight_mode=1),
                dict(algo=9, domain=0, order=3, weight_dtype=1)):
        with pytest.raises(ra.RsrlHipError):                     # (the order-7 wave family runs both since round 5, with bf16 weights since round 6: tests/test_gpu_wave_aux.py)
            ra.Context(n_envs=8, policy=1, **bad)


# ------------------------------------------------------------------------------------------------------------------------------
# TD / TDLambda on the generic Fourier orders (any order without a register-family kernel): rsrl_amd/csrc/kernels_td.hpp k_td_mem.
# The reference's prediction agents are generic over the approximator (prediction/td/td.rs:25-59, td_lambda.rs:25-78).
@pytest.mark.parametrize("algo,trace,domain,order", [(7, 0, 0, 6), (7, 0, 1, 2), (8, 0, 0, 7), (8, 1, 2, 2), (8, 2, 0, 6)])
def test_td_generic_fourier_train_bitwise(ra, orc, algo, trace, domain, order):
    # TDLambda steps with the raw TD error (td_lambda.rs:59-62: no learning rate), which grows geometrically on a dense basis of many
    # features: a short run and short episodes keep every number finite, so that the comparison is of numbers
    N, K, cap = (40, 96, 19) if algo == 7 else (40, 12, 5)
    kw = dict(gamma=0.9, lr=0.01, alpha=0.05, lam=0.3)
    ag = orc.make_agent(domain=domain, order=order, algo=algo, policy=orc.RANDOM, seed=13, trace=trace, max_episode_steps=cap, env_offset=3, **kw)
    run = orc.Run(ag, N, "f32d"); run.reset()
    ost = run.train(K)
    with ra.Context(domain=domain, order=order, n_envs=N, algo=algo, policy=ra.RANDOM, seed=13, trace=trace, max_episode_steps=cap, env_offset=3, **kw) as c:
        assert c.n_out == 1 and c.F == (order + 1) ** c.D
        c.reset()
        st = [c.train(k) for k in (K // 3, 1, K - K // 3 - 1)]
        assert c.timing_read()[2] == "k_td_mem"
        assert np.array_equal(c.states.T, run.state) and np.array_equal(c.actions, run.action)
        for i in range(N):
            assert np.array_equal(c.get_weights(i).reshape(-1), run.weights[i].reshape(-1)), i
            if algo == 8:
                assert np.array_equal(c.get_traces(i).reshape(-1), run.traces[i].reshape(-1)), i
        assert np.abs(run.weights).max() > 0 and np.all(np.isfinite(run.weights))
        assert sum(s["episodes"] for s in st) == ost["episodes"] > 0
        assert sum(s["episodes_truncated"] for s in st) == ost["episodes_truncated"]
        assert abs(sum(s["sum_abs_td_error"] for s in st) - ost["sum_abs_td_error"]) <= 1e-6 * ost["sum_abs_td_error"]


@pytest.mark.parametrize("algo,trace", [(7, 0), (8, 0), (8, 1), (8, 2)])
def test_td_generic_fourier_handle_and_evaluate_bitwise(ra, orc, algo, trace):
    M, domain, order = 48, 1, 2
    rng = np.random.default_rng(algo * 5 + trace)
    kw = dict(gamma=0.97, lr=0.02, alpha=0.1, lam=0.9)
    ag = orc.make_agent(domain=domain, order=order, algo=algo, policy=orc.RANDOM, seed=4, trace=trace, **kw)
    lo, hi = orc.domain_bounds(domain)
    s = (lo[:, None] + (hi - lo)[:, None] * rng.random((len(lo), M))).astype(np.float32) * 0.5
    a = rng.integers(0, 2, M).astype(np.int32)
    with ra.Context(domain=domain, order=order, n_envs=M, algo=algo, policy=ra.RANDOM, seed=4, trace=trace, **kw) as c:
        F = c.F
        c.states = s
        frm, nxt, rew, term = c.domain_step(a)
        term[::7] = 1
        Ws = (rng.normal(size=(M, F)) * 0.1).astype(np.float32)
        Zs = (rng.normal(size=(M, F)) * 0.4).astype(np.float32)
        for i in range(M):
            c.set_weights(Ws[i].reshape(F, 1), i)
            if algo == 8:
                c.set_traces(Zs[i].reshape(F, 1), i)
        v = c.q_evaluate(s)
        assert v.shape == (1, M)
        td = c.handle(frm, a, rew, nxt, term)
        for i in range(M):
            assert v[0, i] == np.float32(orc.v_evaluate(ag, Ws[i], s[:, i], "f32d")), i
            W, Z = Ws[i].copy(), Zs[i].copy()
            d = orc.handle_td(ag, W, Z if algo == 8 else None, frm[:, i], rew[i], nxt[:, i], term[i], "f32d")
            assert td[i] == np.float32(d), (i, td[i], d)
            assert np.array_equal(c.get_weights(i).reshape(-1), W), i
            if algo == 8:
                assert np.array_equal(c.get_traces(i).reshape(-1), Z), i
    with ra.Context(domain=1, order=7, n_envs=8, algo=7, policy=ra.RANDOM) as c:     # the order-7 wave family: prediction kernels since round 5 (tests/test_gpu_wave_aux.py)
        assert c.n_out == 1 and c.F == 4096
    with ra.Context(domain=1, order=7, n_envs=8, algo=7, policy=ra.RANDOM, weight_dtype=ra.W_BF16) as c:      # bf16 weights: round 6
        assert c.n_out == 1
