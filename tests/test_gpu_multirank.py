"""GPU tests of the multi-rank shared-W path on a ONE-GPU box:
  * a communicator of size 1 (RCCL) and a peer group of size 1 run the same finalize -> exchange -> apply sequence as
    world_size > 1 and must reproduce the exchange-free path bit for bit;
  * G ranks as G ctxs with their own streams on one device, driven by G host threads, exchanging the weight delta through
    the one-hop peer-write buffers (the "G streams on one device" simulation of SURVEY Appendix D);
  * 2 ranks as 2 PROCESSES on the one device: the receive buffers are mapped with hipIpc exactly as they are across GPUs,
    the control plane (gloo) all-gathers the handles -- the multi-process code path end to end.
Every replica must hold bit-identical weights (each rank sums the slots in rank order) and agree with the unsharded run up
to the fp32 summation order."""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C4 = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, weight_mode=1, seed=0, max_episode_steps=200)


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


def _run(c, steps=(40, 70)):
    c.reset()
    for k in steps:
        c.train(k, want_stats=False)
    c.sync()
    return c.get_weights(), c.states, c.actions


@pytest.mark.parametrize("kind", ["dense", "tile"])
def test_size1_communicator_runs_the_multirank_sequence_bitwise(ra, kind):
    # (a) no exchange, (b) RCCL communicator of size 1, (c) peer group of size 1: identical weights, states, actions.
    # 110 batch-steps = plain launches + 3 graph replays (the all-reduce / peer kernels are captured with the step)
    N = 8192
    kw = dict(C4, n_envs=N, lr=0.001 / N) if kind == "dense" else \
        dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=1, policy=1, epsilon=0.1, gamma=0.99, weight_mode=1, seed=0,
             max_episode_steps=200, n_envs=N, lr=0.1 / 8 / N)
    with ra.Context(**kw) as plain, ra.Context(**kw) as rccl, ra.Context(exchange=ra.EXCHANGE_PEER, **kw) as peer:
        rccl.comm_init(ra.Context.comm_unique_id(), 1, 0)
        peer.peer_connect([peer.peer_export(1)], 0)
        ref = _run(plain)
        assert np.abs(ref[0]).max() > 0
        for other in (rccl, peer):
            got = _run(other)
            # (tile coding included: every cross-learner sum is 64-bit fixed point -- exact whatever order the atomics retire in)
            assert all(np.array_equal(a, b) for a, b in zip(ref, got))
        # rsrl_hip_handle all-reduces too: a teacher-forced mini-batch gives the same W with and without a communicator
        s = plain.states
        a = plain.actions
        for c in (plain, rccl, peer):
            c.states = s
        frm, nxt, rew, term = plain.domain_step(a)
        before = plain.get_weights()
        w = []
        for c in (plain, rccl, peer):
            c.handle(frm, a, rew, nxt, term)
            w.append(c.get_weights())
        # (handle accumulates the mini-batch delta in 64-bit fixed point: exact, whatever order the device atomics retire in)
        step = np.abs(w[0] - before).max()
        assert step > 0
        for other in w[1:]:
            assert np.array_equal(other, w[0])


G_STREAMS = r'''
import os, sys, json, threading
import numpy as np
sys.path.insert(0, os.environ["RSRL_ROOT"])
import rsrl_amd as ra
G, N = int(os.environ["RSRL_G"]), int(os.environ["RSRL_N"])
C4 = json.loads(os.environ["RSRL_KW"])
from rsrl_amd.distributed import shard_range
def _run(c, steps):
    c.reset()
    for k in steps:
        c.train(k, want_stats=False)
    c.sync()
    return c.get_weights(), c.states, c.actions
kw = dict(C4, exchange=ra.EXCHANGE_PEER)
kw.setdefault("lr", 0.001 / N)
ctxs = [ra.Context(n_envs=cnt, env_offset=off, **kw) for off, cnt in (shard_range(N, G, r) for r in range(G))]
handles = [c.peer_export(G) for c in ctxs]
for r, c in enumerate(ctxs):
    c.peer_connect(handles, r)
out, errs = [None] * G, []
def work(r):
    try:
        out[r] = _run(ctxs[r], (30, 45))
    except Exception as e:
        errs.append(repr(e))
th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
[t.start() for t in th]
[t.join(120) for t in th]
assert not errs and all(o is not None for o in out), errs
for r in range(1, G):
    assert np.array_equal(out[0][0], out[r][0]), "replicas of W diverged"
with ra.Context(n_envs=N, **dict(kw, exchange=ra.EXCHANGE_RCCL)) as full:
    ref = _run(full, (30, 45))
err_w = float(np.max(np.abs(ref[0] - out[0][0])) / max(1.0, np.abs(ref[0]).max()))
states = np.concatenate([o[1] for o in out], axis=1)
same = float(np.all(states == ref[1], axis=0).mean())
print("RESULT " + json.dumps({"err_w": err_w, "same": same, "absw": float(np.abs(ref[0]).max())}), flush=True)
os._exit(0)
'''


@pytest.mark.parametrize("G,N", [(4, 4096), (8, 4096), (8, 4000), (8, 8 * 512 + 77)])
def test_g_ranks_as_g_streams_on_one_device(ra, tmp_path, G, N):
    # G ctxs = G ranks on ONE device in one process, one host thread each, peer-write exchange through same-process pointers: the real world
    # size of a node (8: hop-2 fans out to 8 receive buffers), shards that are not whole 512-learner blocks (4000 / 8 = 500) and a ragged split
    # (4173 = 5 x 522 + 3 x 521).
    # A rank's waiting kernel must not sit in front of a peer's kernels in the same hardware queue, so the process gets more
    # hardware queues than ranks (GPU_MAX_HW_QUEUES, read by the HIP runtime at start-up: hence the subprocess).
    script = tmp_path / "gstreams.py"
    script.write_text(G_STREAMS)
    env = dict(os.environ, RSRL_ROOT=ROOT, RSRL_KW=json.dumps(C4), GPU_MAX_HW_QUEUES=str(2 * G), RSRL_G=str(G), RSRL_N=str(N))
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "RESULT " in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    # same mini-batch rule.  A 512-learner block sums its learners' terms in fp32 (fixed order) before the exact 64-bit fixed-point sums across
    # blocks and ranks take over: shards that are whole blocks keep every block's membership, hence the same bits as the unsharded run; other
    # splits regroup the learners into different blocks and agree to the rounding of those block sums
    if N % (G * 512) == 0:
        assert d["absw"] > 0 and d["err_w"] == 0.0 and d["same"] == 1.0, d
    else:
        assert d["absw"] > 0 and 0 < d["err_w"] <= 1e-6 and d["same"] >= 0.99, d       # measured 1.5e-7 (relative to max(1, |W|))


@pytest.mark.parametrize("kind", ["tile", "sparse_lambda"])
def test_g_ranks_shared_tile_table(ra, tmp_path, kind):
    # the same four in-process ranks over ONE shared tile-coded table: ExpectedSARSA (C3's agent), and SARSA(lambda) with sparse per-learner traces
    # (round 5).  Every rank sums its learners' terms exactly (64-bit fixed point), converts once, and the ranks' float deltas are added in rank order:
    # replicas identical; against the unsharded run the G-term float sum regroups -- equal to its rounding.
    G, N = 4, 2048
    kw = dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, policy=1, epsilon=0.1, gamma=0.99, weight_mode=1, seed=0, max_episode_steps=200)
    kw.update(dict(algo=1, lr=0.1 / 8 / N) if kind == "tile" else dict(algo=3, alpha=0.1 / 8 / N, lam=0.9, trace=0))
    script = tmp_path / "gstreams.py"
    script.write_text(G_STREAMS)
    env = dict(os.environ, RSRL_ROOT=ROOT, RSRL_KW=json.dumps(kw), GPU_MAX_HW_QUEUES=str(2 * G), RSRL_G=str(G), RSRL_N=str(N))
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "RESULT " in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert d["absw"] > 1e-3 and d["err_w"] <= 2e-9 and d["same"] == 1.0, d          # measured 1.2e-10 / 2.3e-10 absolute at |W| = 1.5e-3


def test_missing_peer_times_out_instead_of_hanging(ra):
    # rank 1 never steps: rank 0's exchange gives up after its bounded spin and the next sync reports it
    # (config.peer_timeout_ms: 300 ms here instead of the default 4 s); the update of the failed step is NOT applied, and every
    # synchronising read reports the failure, not only rsrl_hip_sync
    import time
    kw = dict(C4, lr=1e-6, exchange=ra.EXCHANGE_PEER, peer_timeout_ms=300)
    a, b = ra.Context(n_envs=256, **kw), ra.Context(n_envs=256, env_offset=256, **kw)
    h = [a.peer_export(2), b.peer_export(2)]
    a.peer_connect(h, 0); b.peer_connect(h, 1)
    a.reset()
    t0 = time.perf_counter()
    a.train(3, want_stats=False)
    with pytest.raises(ra.RsrlHipError) as ei:
        a.sync()
    assert ei.value.code == -4 and "timed out" in str(ei.value)
    assert time.perf_counter() - t0 < 3.0                      # the configured bound, not the default
    for read in (a.get_weights, lambda: a.states, a.checksum):
        with pytest.raises(ra.RsrlHipError) as ei:
            read()
        assert ei.value.code == -4
    a.close(); b.close()
    with pytest.raises(ra.RsrlHipError):
        ra.Context(n_envs=4, peer_timeout_ms=-1)


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["RSRL_ROOT"])
import rsrl_amd
from rsrl_amd.distributed import ControlPlane, make_sharded_context
cp = ControlPlane()
N = int(os.environ["RSRL_TOTAL"])
kw = json.loads(os.environ["RSRL_KW"])
import atexit, traceback
def _die(*a):
    traceback.print_exc(); sys.stderr.flush(); os._exit(3)      # never linger in a collective the peer will not join
sys.excepthook = lambda *a: (traceback.print_exception(*a), os._exit(3))
ctx = make_sharded_context(N, cp, device=0, **kw)
ctx.reset()
cp.barrier()
for k in (30, 45):
    ctx.train(k, want_stats=False)
ctx.sync()
w = ctx.get_weights()
td = ctx.handle(ctx.states, ctx.actions, np.zeros(ctx.N, np.float32), ctx.states, np.ones(ctx.N, np.uint8))
w2 = ctx.get_weights()
print("RESULT " + json.dumps({"rank": cp.rank, "w": w.tolist(), "w2": w2.tolist(), "states": ctx.states.tolist(), "chk": list(ctx.checksum())}), flush=True)
os._exit(0)
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_processes_one_gpu_peer_exchange_over_hipipc(ra, tmp_path):
    N = 4096
    kw = dict(C4, lr=0.001 / N, exchange=1)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RSRL_ROOT=ROOT, RSRL_TOTAL=str(N), RSRL_KW=json.dumps(kw), GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res, outs = {}, []
    for p in procs:
        try:
            so, se = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
        outs.append((p.returncode, so, se[-2000:]))
    if any("timed out" in o[2] for o in outs) and os.environ.get("RSRL_ALLOW_TIMESLICE_SKIP") == "1":
        # kernels of different PROCESSES did not run at the same time on this box's single GPU (exclusive time slicing): a
        # rank's bounded wait cannot overlap the peer's push.  That is a property of sharing ONE device, not of the exchange
        # (one process per GPU is the deployment); the in-process G-streams test above covers the exchange itself.
        # An explicit opt-in only: by default a time-out FAILS below, so a regression cannot turn green by skipping.
        pytest.skip("processes are time-sliced exclusively on this GPU: " + repr([o[2][-200:] for o in outs]))
    for rc, so, se in outs:
        assert rc == 0, (so, se)
        d = json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][0][7:])
        res[d["rank"]] = d
    w0, w1 = np.array(res[0]["w"], np.float32), np.array(res[1]["w"], np.float32)
    assert np.array_equal(w0, w1) and np.abs(w0).max() > 0, "replicas of W diverged across processes"
    assert np.array_equal(np.array(res[0]["w2"], np.float32), np.array(res[1]["w2"], np.float32))     # handle() exchanged too
    assert res[0]["chk"][0] == res[1]["chk"][0]
    with ra.Context(n_envs=N, **dict(kw, exchange=0)) as full:
        ref = _run(full, steps=(30, 45))
    assert np.max(np.abs(ref[0] - w0)) <= 2e-6 * max(1.0, np.abs(ref[0]).max())
    states = np.concatenate([np.array(res[r]["states"], np.float32) for r in (0, 1)], axis=1)
    assert np.all(np.abs(states - ref[1]) <= 1e-6, axis=0).mean() >= 0.99
