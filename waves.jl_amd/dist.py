"""Episode-level data parallelism over the GPUs of one node.

The reference is single-GPU; independent episodes / environments shard embarrassingly, so the only communication is
one broadcast of the design-space block from rank 0 at start (RCCL over xGMI when the backend is "nccl") and one gather
of the energy traces at the end.  No collective sits in the time loop and no halo is ever exchanged.
One process per GPU; rendezvous through the usual RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* variables.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np

from .designs import AdjustableRadiiScatterers, Cloak, Cylinders, DesignSpace


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shares_device(world: int, local_rank: int, ndev: int) -> bool:
    """True when the ranks of this node cannot each have a GPU of their own (ndev = visible devices, 0 = CPU-only run)."""
    return ndev > 0 and (world > ndev or local_rank >= ndev)


def init(backend: str | None = None):
    """Initialise torch.distributed if WORLD_SIZE > 1.  backend None -> "nccl" (= RCCL) when a GPU is visible, else gloo."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world == 1 and not os.environ.get("WAVES_AMD_FORCE_DIST"):  # (the variable lets a 1-GPU box exercise the RCCL path)
        return rank, local_rank, world
    ndev = torch.cuda.device_count()
    if shares_device(world, local_rank, ndev):
        # Ranks that share a GPU (rehearsals on a one-GPU box, WAVES_AMD_ALLOW_SHARED_GPU=1) must not use the resident step
        # kernel: it needs ALL its tiles on the device at once, the library counts contexts per PROCESS, and two processes'
        # resident grids can each end up partly resident -- every tile then polls in vain until the launch gives up.
        # Decided here, before any Context exists, for every caller (bench.py, tools/rollout.py, user scripts).
        os.environ["WAVES_AMD_FUSED_RESIDENT"] = "0"
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        # a rehearsal with several ranks on ONE GPU (WAVES_AMD_ALLOW_SHARED_GPU=1): RCCL refuses two ranks on a device
        if backend == "nccl" and os.environ.get("WAVES_AMD_ALLOW_SHARED_GPU") and ndev < world:
            backend = "gloo"
    if backend == "nccl":
        n = torch.cuda.device_count()
        if n == 0:
            raise RuntimeError("waves_jl_amd.dist.init: backend nccl (RCCL) needs a GPU")
        # (ranks that share a device -- rehearsals on a one-GPU box -- are the caller's decision: see bench.py)
        torch.cuda.set_device(local_rank % n)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _device():
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def shard_episodes(n_episodes: int, world: int, rank: int) -> range:
    """rank r takes the contiguous block of episodes [r*E/W, (r+1)*E/W) (config 3: 64 -> 8 per GPU)."""
    base, rem = divmod(n_episodes, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


# --- the design-space block: [M_config, M_core] + low/high (pos, r, c) of config and core, fp32 -------------------
def pack_design_space(ds: DesignSpace) -> np.ndarray:
    lo, hi = ds.low, ds.high
    if not isinstance(lo, Cloak):
        raise TypeError("pack_design_space: Cloak design spaces (build_*_design_space) are what WaveEnv uses")
    parts = [np.array([len(lo.config), len(lo.core)], np.float32)]
    for d in (lo, hi):
        for cyl in (d.config.cylinders, d.core):
            parts += [cyl.pos.ravel(order="F"), cyl.r, cyl.c]
    return np.concatenate(parts).astype(np.float32)


def unpack_design_space(buf: np.ndarray) -> DesignSpace:
    buf = np.asarray(buf, np.float32)
    mc, mk = int(buf[0]), int(buf[1])
    off = 2
    out = []
    for _ in range(2):
        cyls = []
        for m in (mc, mk):
            pos = buf[off:off + 2 * m].reshape(m, 2, order="F"); off += 2 * m
            r = buf[off:off + m]; off += m
            c = buf[off:off + m]; off += m
            cyls.append(Cylinders(pos, r, c))
        out.append(Cloak(AdjustableRadiiScatterers(cyls[0]), cyls[1]))
    return DesignSpace(out[0], out[1])


def broadcast_design_space(ds: DesignSpace | None, src: int = 0) -> DesignSpace:
    """Every rank ends up with rank `src`'s design space (one small broadcast; latency-bound)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return ds
    dev = _device()
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    payload = None
    if dist.get_rank() == src:
        payload = torch.from_numpy(pack_design_space(ds)).to(dev)
        n[0] = payload.numel()
    dist.broadcast(n, src=src)
    if payload is None:
        payload = torch.empty(int(n.item()), dtype=torch.float32, device=dev)
    dist.broadcast(payload, src=src)
    return unpack_design_space(payload.cpu().numpy())


def broadcast_field(field: np.ndarray | None, shape, src: int = 0) -> np.ndarray:
    """Broadcast one (nx, ny) fp32 field (e.g. a design/wave-speed field or a source shape) from rank `src`."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return field
    dev = _device()
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(field, np.float32).reshape(-1)).to(dev)
    else:
        t = torch.empty(int(np.prod(shape)), dtype=torch.float32, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().reshape(shape)


def gather_signals(sig: np.ndarray) -> List[np.ndarray] | None:
    """all_gather of each rank's stacked energy traces (same shape on every rank); returns the list on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [np.asarray(sig)]
    dev = _device()
    t = torch.from_numpy(np.ascontiguousarray(sig, np.float32)).to(dev)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]


def max_over_ranks(x: float) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()


def finalize():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
