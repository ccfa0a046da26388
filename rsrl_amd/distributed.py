"""Multi-GPU host logic: one process per GPU (launched by torch.distributed.run), environments sharded
contiguously by GLOBAL id.  Per-env weights need no data-path collective at all; shared weights all-reduce the
(F x A) weight delta once per batch-step over RCCL inside librsrl_hip.so (rsrl_hip_comm_*).  torch.distributed
(gloo) is only the control plane here: rendezvous, the ncclUniqueId broadcast, barriers and the max-over-ranks
of the timings."""
import os
from dataclasses import dataclass


def shard_range(n_total, world, rank):
    """Contiguous partition of [0, n_total): GPU g owns [offset, offset + count)."""
    if not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


@dataclass
class RankInfo:
    rank: int = 0
    local_rank: int = 0
    world: int = 1

    @staticmethod
    def from_env(env=os.environ):
        return RankInfo(int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1")))


class ControlPlane:
    """Thin wrapper over a torch.distributed (gloo) group; a no-op for world == 1."""

    def __init__(self, info=None, backend="gloo"):
        self.info = info or RankInfo.from_env()
        self._dist = None
        if self.info.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, init_method="env://", rank=self.info.rank,
                                        world_size=self.info.world)
            self._dist = dist

    @property
    def rank(self):
        return self.info.rank

    @property
    def world(self):
        return self.info.world

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def broadcast_bytes(self, payload, src=0):
        """payload (bytes) on `src`, None elsewhere -> bytes on every rank"""
        if self._dist is None:
            return payload
        box = [payload if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def all_gather_bytes(self, payload):
        """every rank's payload (bytes), in rank order, on every rank"""
        if self._dist is None:
            return [payload]
        out = [None] * self.world
        self._dist.all_gather_object(out, payload)
        return out

    def max_over_ranks(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, array):
        """in-place sum of a numpy float64 array over the ranks (control-plane sized data only)"""
        if self._dist is None:
            return array
        import torch
        t = torch.from_numpy(array)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return array

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def host_identity():
    """what tells two ranks they run on the SAME machine: the kernel's boot id (unique per boot, shared by every container of the host) plus
    the hostname"""
    import socket
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        boot = ""
    return f"{boot}:{socket.gethostname()}"


def rank_topology(device, device_identity=None, device_count=None, can_access_peer=None):
    """what one rank tells the others for the RSRL_EXCHANGE_AUTO decision: its host, the physical identity of its device, and which physical
    devices it sees and can reach from its own (identity -> bool).  Ordinals mean nothing across processes (HIP_VISIBLE_DEVICES)."""
    if device_identity is None:
        from .context import can_access_peer, device_count, device_identity
    n = device_count()
    return {"host": host_identity(), "device": device_identity(device),
            "reach": {device_identity(d): bool(can_access_peer(device, d)) for d in range(n)}}


def choose_exchange(topologies):
    """RSRL_EXCHANGE_AUTO, decided identically by every rank from the all-gathered rank_topology records (None = that rank could not tell):
    the one-hop peer exchange (1) only if ALL ranks run on one host and every rank sees AND can access every other rank's physical device --
    a rank on another node, a device hidden from a peer by HIP_VISIBLE_DEVICES, or a pair without peer access makes it RCCL (0), which
    works on any topology."""
    if not topologies or any(t is None for t in topologies):
        return 0
    if len({t["host"] for t in topologies}) != 1:
        return 0
    for t in topologies:
        for o in topologies:
            if not t["reach"].get(o["device"], False):
                return 0
    return 1


def make_sharded_context(total_envs, control, context_cls=None, unique_id_fn=None, force_exchange=False, **cfg):
    """Create this rank's Context for its shard of `total_envs` environments.  In shared-weight mode with more
    than one rank the exchange of the weight delta is set up: exchange=EXCHANGE_AUTO (default) takes the one-hop peer
    exchange when all ranks share one host and every rank sees and can access every other rank's physical device (choose_exchange), RCCL otherwise; EXCHANGE_PEER / _RCCL
    force one.  force_exchange attaches it for a single rank too (a group of size 1 runs the same sequence: how the
    multi-rank path is exercised on a one-GPU box).

    FAILS TOGETHER: a rank whose local step fails (ctx creation, export, connect) still joins every control-plane
    collective of the set-up with a marker, and then every rank raises -- no rank is left waiting in a collective
    (an all-gather, ncclCommInitRank) for a peer that has already given up."""
    if context_cls is None:
        from .context import Context as context_cls
    if total_envs < control.world:
        raise ValueError(f"{total_envs} environments cannot be sharded over {control.world} ranks (every rank needs at least one)")
    offset, count = shard_range(total_envs, control.world, control.rank)
    cfg = dict(cfg)
    if "device" not in cfg:
        from .context import device_count
        cfg["device"] = control.info.local_rank % max(1, device_count()) if context_cls.__name__ == "Context" else control.info.local_rank
    shared = cfg.get("weight_mode", 0) == 1
    attach = shared and (control.world > 1 or force_exchange)
    err, ctx = None, None
    try:
        ctx = context_cls(n_envs=count, env_offset=cfg.pop("env_offset", 0) + offset, **cfg)
    except Exception as e:      # noqa: BLE001
        if not attach:
            raise
        err = e

    def together(what):
        """every rank learns whether any rank failed so far; all raise together"""
        oks = control.all_gather_bytes(err is None)
        if not all(oks):
            bad = [r for r, ok in enumerate(oks) if not ok]
            if ctx is not None and hasattr(ctx, "close"):
                ctx.close()
            raise err if err is not None else RuntimeError(f"shared-W exchange set-up ({what}) failed on rank(s) {bad}; this rank gives up with them")

    if attach:
        exchange = cfg.get("exchange", 2)
        if exchange == 2:                      # AUTO: one decision from the all-gathered (host, physical device, reachable devices) of every rank
            mine = None
            if err is None and context_cls.__name__ == "Context":
                try:
                    mine = rank_topology(cfg["device"])
                except Exception:      # noqa: BLE001
                    mine = None
            elif err is None:
                mine = getattr(context_cls, "fake_topology", lambda d: None)(cfg["device"])
            exchange = choose_exchange(control.all_gather_bytes(mine))
        together("creating the ctxs")
        if exchange == 1:                      # one-hop peer-write: all-gather the receive-buffer handles
            h = None
            try:
                h = ctx.peer_export(control.world)
            except Exception as e:      # noqa: BLE001
                err = e
            handles = control.all_gather_bytes(h)
            together("exporting the receive buffers")
            try:
                ctx.peer_connect(handles, control.rank)
            except Exception as e:      # noqa: BLE001
                err = e
            together("connecting the receive buffers")
        else:                                  # RCCL: rank 0 draws the ncclUniqueId, the control plane broadcasts it
            uid = None
            if control.rank == 0:
                try:
                    uid = (unique_id_fn or context_cls.comm_unique_id)()
                except Exception as e:      # noqa: BLE001
                    err = e
            uid = control.broadcast_bytes(uid, src=0)
            together("drawing the communicator id")          # (ncclCommInitRank blocks until every rank joins: nobody enters it alone)
            try:
                ctx.comm_init(uid, control.world, control.rank)
            except Exception as e:      # noqa: BLE001
                err = e
            together("initialising the communicator")
    return ctx
