"""The N > 1 RCCL path of the shared-W exchange, executed on ONE device through a test double of RCCL (tests/stubs/rccl_stub.cpp).

Real RCCL refuses ranks that share a device and the box has one GPU, so rsrl_hip_group_create(RCCL) / rsrl_hip_group_train / comm_init with more
than one rank had never run in any form (VERDICT r4 missing #3).  LD_PRELOADing the double -- it really all-reduces the ranks' buffers, in rank
order, and records every call -- lets G = 8 ranks run end to end and lets the test assert the CALL PROTOCOL a real RCCL needs:

* single-thread group (rsrl_hip_group_train): per batch-step ONE ncclGroupStart / ncclGroupEnd pair holding exactly one in-place all-reduce per
  rank (ncclInt64 table for the dense basis, ncclFloat delta for tile coding), nothing un-grouped after the warm-up;
* one thread per rank (comm_init + rsrl_hip_train): un-grouped all-reduces, one per rank and batch-step, that rendezvous;
* both: every replica of W identical, and -- shards of whole 512-learner blocks, the sums across ranks exact 64-bit integers -- identical to the
  unsharded run bit for bit (dense basis).
What stays hardware-only: RCCL's own ring / xGMI transport."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes, json, os, sys, threading
import numpy as np
sys.path.insert(0, os.environ["RSRL_ROOT"])
import rsrl_amd as ra
stub = ctypes.CDLL(os.environ["RCCL_STUB"])
def log():
    n = stub.rccl_stub_log(None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    stub.rccl_stub_log(buf, n + 1)
    return buf.value.decode().splitlines()
G, mode, kind = int(os.environ["G"]), os.environ["MODE"], os.environ["KIND"]
N = 4000 if os.environ.get("RAGGED") else 4096                      # ragged: shards of 500 learners, not whole 512-learner blocks
if kind == "dense":
    kw = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, weight_mode=1, seed=0, max_episode_steps=200, lr=0.001 / N, exchange=ra.EXCHANGE_RCCL)
elif kind == "tile":
    kw = dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=1, policy=1, epsilon=0.1, gamma=0.99, weight_mode=1, seed=0, max_episode_steps=200,
              lr=0.1 / 8 / N, exchange=ra.EXCHANGE_RCCL)
else:                                                                 # SARSALambda / QLambda over the shared table: sparse per-learner traces (ADVICE r5)
    kw = dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=3 if kind == "sarsa_lambda" else 4, policy=1, epsilon=0.1, gamma=0.99, weight_mode=1, seed=0,
              max_episode_steps=200, alpha=0.1 / 8 / N, lam=0.9, trace=0, exchange=ra.EXCHANGE_RCCL)
from rsrl_amd.distributed import shard_range
shards = [shard_range(N, G, r) for r in range(G)]
ctxs = [ra.Context(n_envs=cnt, env_offset=off, **kw) for off, cnt in shards]
K = (7, 12)
out = {}
if mode == "group":                                                   # ONE host thread for all ranks
    ra.Context.group_create(ctxs)
    for c in ctxs: c.reset()
    stub.rccl_stub_log_clear()
    for k in K: ra.Context.group_train(ctxs, k)
    for c in ctxs: c.sync()
    out["log"] = log()
    try:
        ctxs[0].train(1)
        out["train_on_group_rank"] = "accepted"
    except ra.RsrlHipError as e:
        out["train_on_group_rank"] = e.code
else:                                                                 # one host thread per rank, as one process per GPU would
    uid = ra.Context.comm_unique_id()
    errs = []
    def work(r):
        try:
            ctxs[r].comm_init(uid, G, r)
            ctxs[r].reset()
            for k in K: ctxs[r].train(k, want_stats=False)
            ctxs[r].sync()
        except Exception as e:
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]; [t.join(180) for t in th]
    assert not errs, errs
    out["log"] = log()
out["info"] = [list(c.comm_info()) for c in ctxs]
W = [c.get_weights() for c in ctxs]
out["replicas_equal"] = bool(all(np.array_equal(W[0], w) for w in W[1:]))
with ra.Context(n_envs=N, **dict(kw, exchange=ra.EXCHANGE_AUTO)) as full:
    full.reset()
    for k in K: full.train(k, want_stats=False)
    Wf, Sf = full.get_weights(), full.states
out["absw"] = float(np.abs(Wf).max())
out["err_w"] = float(np.abs(W[0] - Wf).max())
S = np.concatenate([c.states for c in ctxs], axis=1)
out["states_same"] = float(np.all(S == Sf, axis=0).mean())
print("RESULT " + json.dumps(out), flush=True)
os._exit(0)
'''


def _run(tmp_path, G, mode, kind, ragged=False):
    from rsrl_amd import _build
    stub = _build.build_rccl_stub()
    script = tmp_path / "rccl_stub_run.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, RSRL_ROOT=ROOT, RCCL_STUB=stub, LD_PRELOAD=stub, G=str(G), MODE=mode, KIND=kind, RSRL_NO_GRAPH="1",
               GPU_MAX_HW_QUEUES=str(2 * G))
    if ragged:
        env["RAGGED"] = "1"
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RESULT " in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])


@pytest.mark.parametrize("kind,ragged", [("dense", False), ("dense", True), ("tile", False), ("sarsa_lambda", False), ("q_lambda", True)])
def test_single_thread_group_of_8_ranks_groups_every_all_reduce(tmp_path, kind, ragged):
    G, steps = 8, 19
    d = _run(tmp_path, G, "group", kind, ragged)
    assert d["info"] == [[G, r, 0] for r in range(G)]                  # what the communicator itself reports: world 8, rank r, RCCL
    assert d["replicas_equal"] and d["absw"] > 0
    if kind == "dense" and not ragged:
        assert d["err_w"] == 0.0 and d["states_same"] == 1.0, d       # whole 512-learner blocks per rank + exact 64-bit sums across ranks: sharded == unsharded
    elif kind == "dense":
        assert d["err_w"] <= 1e-6 * max(1.0, d["absw"]) and d["states_same"] >= 0.99, d   # other splits regroup the fp32 block sums (test_gpu_multirank)
    else:                                                              # (the lambda agents over the shared table step their sparse-trace kernels in lock-step:
        assert d["err_w"] <= 2e-6 * max(1.0, d["absw"]) and d["states_same"] >= 0.99, d   # the single-step kernel's TD rule would learn something else)
        # float delta: the summation order over ranks differs
    # ---- the protocol: per batch-step one group with one in-place all-reduce per rank, ranks 0..G-1, and nothing outside a group
    log = [ln for ln in d["log"] if not ln.startswith("CommDestroy")]
    assert log.count("GroupStart") == steps and sum(ln.startswith("GroupEnd rc=0") for ln in log) == steps, log[:40]
    i = 0
    for _ in range(steps):
        assert log[i] == "GroupStart", (i, log[i])
        body = log[i + 1:i + 1 + G]
        want_type = 4 if kind == "dense" else 7                        # ncclInt64 / ncclFloat
        for r, ln in enumerate(body):
            assert ln.startswith(f"AllReduce rank={r} ") and f"type={want_type} " in ln and "inplace=1 grouped=1" in ln, ln
        assert len({ln.split("count=")[1].split()[0] for ln in body}) == 1
        assert log[i + 1 + G] == "GroupEnd rc=0"
        i += G + 2
    assert i == len(log), log[i:i + 5]
    assert d["train_on_group_rank"] == -5                              # rsrl_hip_train on a rank of a single-thread RCCL group: ESTATE


def test_one_thread_per_rank_all_reduces_rendezvous(tmp_path):
    G, steps = 4, 19
    d = _run(tmp_path, G, "threads", "dense")
    assert d["info"] == [[G, r, 0] for r in range(G)] and d["replicas_equal"] and d["err_w"] == 0.0 and d["states_same"] == 1.0, d
    ar = [ln for ln in d["log"] if ln.startswith("AllReduce")]
    assert all("grouped=0" in ln and "inplace=1" in ln for ln in ar)
    # the warm-up of comm_init (ncclFloat) + one table all-reduce (ncclInt64) per batch-step, per rank
    for r in range(G):
        mine = [ln for ln in ar if f"rank={r} " in ln]
        assert sum("type=7 " in ln for ln in mine) == 1 and sum("type=4 " in ln for ln in mine) == steps, (r, len(mine))
