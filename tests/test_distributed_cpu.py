"""world_size-2 gloo tests (CPU): the multi-GPU host logic -- contiguous sharding by global env id, the control
plane (unique-id broadcast, max over ranks) -- and the sharding scheme itself checked with the oracle: shards
keyed by GLOBAL env id reproduce the unsharded run (per-env weights: exactly; shared weights: the per-step
delta all-reduce gives the unsharded update)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["RSRL_ROOT"])
from rsrl_amd.distributed import ControlPlane, RankInfo, shard_range, make_sharded_context
from oracle import oracle as orc

cp = ControlPlane()
out = {"rank": cp.rank, "world": cp.world}
# control plane
uid = cp.broadcast_bytes(bytes(range(128)) if cp.rank == 0 else None)
out["uid_ok"] = uid == bytes(range(128))
out["max"] = cp.max_over_ranks(1.5 + cp.rank)

# make_sharded_context with a recording stand-in for the device Context (no GPU here)
class FakeCtx:
    calls = []
    def __init__(self, **kw): self.kw = kw
    @staticmethod
    def comm_unique_id(): return b"U" * 128
    def comm_init(self, uid, world, rank): FakeCtx.calls.append((uid, world, rank))
ctx = make_sharded_context(1001, cp, context_cls=FakeCtx, weight_mode=1, seed=3)
out["shard"] = [ctx.kw["env_offset"], ctx.kw["n_envs"], ctx.kw["device"]]
out["comm"] = [(u == b"U" * 128, w, r) for (u, w, r) in FakeCtx.calls]

# a rank whose LOCAL set-up step fails must not leave the other one waiting in a collective: both raise, together
class HalfBroken(FakeCtx):
    def peer_export(self, world):
        if self.kw["env_offset"] > 0:
            raise RuntimeError("export failed on this rank")
        return b"H" * 128
    def peer_connect(self, handles, rank): raise AssertionError("never reached: a peer could not export")
    def close(self): pass
try:
    make_sharded_context(64, cp, context_cls=HalfBroken, weight_mode=1, exchange=1)
    out["fails_together"] = "no error"
except RuntimeError as e:
    out["fails_together"] = str(e)
class PeerOk(FakeCtx):
    def peer_export(self, world): return bytes([self.kw["env_offset"] % 251]) * 128
    def peer_connect(self, handles, rank): self.connected = (handles, rank)
pc = make_sharded_context(64, cp, context_cls=PeerOk, weight_mode=1, exchange=1)
out["peer_connected"] = [len(pc.connected[0]), pc.connected[1], pc.connected[0][1][0]]

# EXCHANGE_AUTO is decided from (host, physical device, reachable devices) of every rank, not from local ordinals: a control plane that reports
# two hosts -- or a device a peer cannot see -- must end up on RCCL; one host with full reach takes the peer exchange
class TopoCtx(PeerOk):
    topo = None
    used = None
    @staticmethod
    def fake_topology(device): return TopoCtx.topo
    def comm_init(self, uid, world, rank): TopoCtx.used = "rccl"
    def peer_connect(self, handles, rank): TopoCtx.used = "peer"
both = {100: True, 101: True}
scen = {"two_hosts": {"host": "A" if cp.rank == 0 else "B", "device": 100, "reach": {100: True}},          # per-rank HIP_VISIBLE_DEVICES on 2 nodes: every rank "device 0"
        "hidden_device": {"host": "A", "device": 100 + cp.rank, "reach": {100 + cp.rank: True}},          # one node, each rank sees only its own GPU
        "no_peer_access": {"host": "A", "device": 100 + cp.rank, "reach": {100 + cp.rank: True, 101 - cp.rank: False}},
        "one_rank_cannot_tell": ({"host": "A", "device": 100 + cp.rank, "reach": both} if cp.rank == 0 else None),
        "one_host_full_reach": {"host": "A", "device": 100 + cp.rank, "reach": both},
        "one_host_shared_device": {"host": "A", "device": 100, "reach": {100: True}}}
out["auto"] = {}
for name, topo in scen.items():
    TopoCtx.topo, TopoCtx.used = topo, None
    make_sharded_context(64, cp, context_cls=TopoCtx, weight_mode=1)
    out["auto"][name] = TopoCtx.used

# the sharding scheme, exercised with the oracle
N, K = 24, 40
off, cnt = shard_range(N, cp.world, cp.rank)
kw = dict(policy=orc.EGREEDY, epsilon=0.1, seed=5, max_episode_steps=15)
run = orc.Run(orc.make_agent(env_offset=off, **kw), cnt, "f64"); run.reset(); st = run.train(K)
out["per_env_state"] = run.state.tolist(); out["per_env_w0"] = run.weights[0].tolist(); out["per_env_eps"] = st["episodes"]
srun = orc.Run(orc.make_agent(env_offset=off, shared_w=True, lr=0.001 / N, **kw), cnt, "f64"); srun.reset()
srun.train_with_dw_hook(K, lambda dW: cp.sum_over_ranks(dW))
out["shared_w"] = srun.weights.tolist(); out["shared_state"] = srun.state.tolist()
cp.barrier()
print("RESULT " + json.dumps(out))
cp.close()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_exactly():
    from rsrl_amd.distributed import shard_range
    for n, w in [(0, 3), (1, 2), (7, 8), (65536, 8), (1001, 2), (1048576, 8), (10, 3)]:
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (o1, c1), (o2, _) in zip(spans, spans[1:]):
            assert o1 + c1 == o2
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_choose_exchange_rules():
    from rsrl_amd.distributed import choose_exchange, host_identity, rank_topology
    t = lambda host, dev, reach: {"host": host, "device": dev, "reach": reach}       # noqa: E731
    assert choose_exchange([t("h", 1, {1: True, 2: True}), t("h", 2, {1: True, 2: True})]) == 1
    assert choose_exchange([t("h", 1, {1: True, 2: True}), t("g", 2, {1: True, 2: True})]) == 0      # two hosts
    assert choose_exchange([t("h", 1, {1: True}), t("h", 2, {1: True, 2: True})]) == 0               # rank 0 does not see device 2
    assert choose_exchange([t("h", 1, {1: True, 2: False}), t("h", 2, {1: True, 2: True})]) == 0     # ... or cannot access it
    assert choose_exchange([t("h", 1, {1: True}), None]) == 0 and choose_exchange([]) == 0
    assert choose_exchange([t("h", 1, {1: True})]) == 1                                              # a group of one
    assert host_identity() == host_identity() and ":" in host_identity()
    # rank_topology through injected device functions: 2 visible devices, the second unreachable
    top = rank_topology(0, device_identity=lambda d: 1000 + d, device_count=lambda: 2, can_access_peer=lambda a, b: a == b)
    assert top["device"] == 1000 and top["reach"] == {1000: True, 1001: False}


def test_sharded_context_needs_one_env_per_rank():
    # a rank with zero environments cannot create its ctx and would leave the others waiting in the communicator set-up
    from rsrl_amd.distributed import ControlPlane, RankInfo, make_sharded_context

    class Never:
        def __init__(self, **kw):
            raise AssertionError("must be refused before any ctx is created")
    cp = ControlPlane(RankInfo(0, 0, 1))
    cp.info.world = 4                       # pretend: 4 ranks, 3 environments
    with pytest.raises(ValueError):
        make_sharded_context(3, cp, context_cls=Never, weight_mode=1)


def test_world2_gloo_sharding_and_control_plane(tmp_path, orc):
    import json
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RSRL_ROOT=ROOT, GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    res = {}
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-2000:]
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][0]
        d = json.loads(line[7:])
        res[d["rank"]] = d
    assert sorted(res) == [0, 1]
    for r in (0, 1):
        assert res[r]["uid_ok"] and res[r]["max"] == 2.5 and res[r]["world"] == 2
        assert res[r]["comm"] == [[True, 2, r]]
        assert res[r]["peer_connected"] == [2, r, 32 % 251]              # both handles, in rank order (rank 1's shard starts at 32)
    # the rank whose export failed raises its own error, the other one gives up WITH it instead of waiting for its handle
    assert "export failed on this rank" in res[1]["fails_together"]
    assert "failed on rank(s) [1]" in res[0]["fails_together"]
    for r in (0, 1):
        pass
    assert res[0]["shard"] == [0, 501, 0] and res[1]["shard"] == [501, 500, 1]
    for r in (0, 1):                                               # the AUTO decision knows hosts and physical devices
        assert res[r]["auto"] == {"two_hosts": "rccl", "hidden_device": "rccl", "no_peer_access": "rccl", "one_rank_cannot_tell": "rccl",
                                  "one_host_full_reach": "peer", "one_host_shared_device": "peer"}, res[r]["auto"]

    # unsharded reference runs
    N, K = 24, 40
    kw = dict(policy=orc.EGREEDY, epsilon=0.1, seed=5, max_episode_steps=15)
    full = orc.Run(orc.make_agent(**kw), N, "f64"); full.reset(); st = full.train(K)
    both = np.concatenate([np.array(res[0]["per_env_state"]), np.array(res[1]["per_env_state"])])
    assert np.array_equal(both, full.state)                       # RNG keyed by GLOBAL env id: sharding is invisible
    assert np.array_equal(np.array(res[1]["per_env_w0"]), full.weights[12])
    assert res[0]["per_env_eps"] + res[1]["per_env_eps"] == st["episodes"]
    sfull = orc.Run(orc.make_agent(shared_w=True, lr=0.001 / N, **kw), N, "f64"); sfull.reset(); sfull.train(K)
    for r in (0, 1):                                              # every rank applied the same summed delta
        assert np.allclose(np.array(res[r]["shared_w"]), sfull.weights, rtol=0, atol=1e-13)
    sboth = np.concatenate([np.array(res[0]["shared_state"]), np.array(res[1]["shared_state"])])
    assert np.allclose(sboth, sfull.state, rtol=0, atol=1e-12)
    assert np.array_equal(np.array(res[0]["shared_w"]), np.array(res[1]["shared_w"]))   # replicas stay bit-identical
