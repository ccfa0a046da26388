#!/usr/bin/env python3
"""Can two PROCESSES make progress on one GPU at the same time?  Rank 0 waits inside a kernel (the peer exchange's bounded
spin) for a delta that rank 1 pushes after a delay; reports how long rank 0's wait took.  If kernels of different processes
ran concurrently the wait ends as soon as rank 1 pushes (~delay); if the processes are time-sliced exclusively it ends at the
spin bound (~4 s).  Usage: torchrun-free -- spawns its two ranks itself."""
import json, os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if "RANK" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill(); so, se = p.communicate()
        print(f"--- rank {r} rc={p.returncode}\n{so[-1500:]}\n{se[-1500:]}")
    sys.exit(0)

import numpy as np
import rsrl_amd
from rsrl_amd.distributed import ControlPlane, make_sharded_context
cp = ControlPlane()
kw = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, weight_mode=1, seed=0, max_episode_steps=200, lr=1e-7, exchange=1)
try:
    ctx = make_sharded_context(1024, cp, device=0, **kw)
    ctx.reset(); ctx.sync()
    cp.barrier()
    delay = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    for k in range(3):
        if cp.rank == 1:
            time.sleep(delay)                      # rank 0 is already waiting inside its kernel
        t0 = time.perf_counter()
        ctx.train(1, want_stats=False)
        try:
            ctx.sync()
            ok = True
        except rsrl_amd.RsrlHipError as e:
            ok = str(e)
        print(json.dumps({"rank": cp.rank, "step": k, "train+sync_s": round(time.perf_counter() - t0, 3), "ok": ok}), flush=True)
        if ok is not True:
            break
    print("W", float(np.abs(ctx.get_weights()).sum()), flush=True)
except Exception as e:      # noqa: BLE001
    print("ERROR", repr(e), flush=True)
os._exit(0)
