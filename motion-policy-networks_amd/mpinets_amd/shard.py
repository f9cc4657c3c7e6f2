"""Sharding of independent planning problems across the GPUs of one node.

Every environment (planning problem) is independent in every stage of the path (SURVEY.md
section 8e), so rank r of W simply owns environments [r*E, (r+1)*E): weights and robot tables are
replicated, there is NO collective on the step, and results come back with one final gather.
``torch.distributed`` (RCCL on the GPU box, gloo in the CPU tests) is used only for the
start/stop barrier of a timed region and for that final gather.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int, int]:
    """Join the job torchrun started (no-op for a single process) and make this rank's GPU the current device --
    for EVERY backend: the engine enqueues on torch's current stream of the current device, so a gloo job that
    skipped ``set_device`` would put every rank's kernels on device 0.  ``device`` defaults to
    ``cuda:LOCAL_RANK``.  Backend: RCCL ("nccl") when GPUs are present, gloo otherwise; ``MPX_DIST_BACKEND``
    overrides (tests run 2 ranks on one GPU with gloo)."""
    rank, ws, local = world()
    if torch.cuda.is_available():
        device = torch.device("cuda", local) if device is None else torch.device(device)
        torch.cuda.set_device(device)
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("MPX_DIST_BACKEND")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=ws, **kwargs)
    return rank, ws, local


def env_range(rank: int, world_size: int, envs_per_rank: int) -> range:
    """Weak scaling: every rank owns ``envs_per_rank`` environments; global ids are contiguous."""
    return range(rank * envs_per_rank, (rank + 1) * envs_per_rank)


def split_even(total: int, world_size: int) -> List[range]:
    """Strong scaling: contiguous, near-equal ranges covering ``total`` environments."""
    base, rem = divmod(total, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < rem else 0)
        out.append(range(start, start + n))
        start += n
    return out


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def _comm_device(device=None):
    """Collectives run on the GPU with RCCL and on the host with gloo."""
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(t: torch.Tensor) -> Optional[torch.Tensor]:
    """Final host gather: equal-shaped per-rank results -> concatenated on rank 0 (None elsewhere)."""
    if not dist.is_initialized():
        return t
    rank, ws = dist.get_rank(), dist.get_world_size()
    t = t.contiguous().to(_comm_device(t.device))
    bufs = [torch.empty_like(t) for _ in range(ws)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    return torch.cat(bufs, dim=0) if rank == 0 else None


# ---- training only (row N1): the one collective of the code base ----------------------------------------
def allreduce_gradients(params, bucket_bytes: int = 64 << 20) -> int:
    """Average ``p.grad`` over the ranks (data-parallel training; the reference uses Lightning's DDP,
    run_training.py:71-77).  Gradients are packed into flat buckets of ``bucket_bytes`` and each bucket is
    one all-reduce: xGMI is point-to-point (a ring all-reduce is bound by one link, ~153 GB/s per direction),
    so a few large messages beat one call per tensor -- the 19 M parameters (76 MB fp32) go out in two
    buckets.  Returns the number of collectives issued (0 for a single process)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    ws = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    calls, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (j == i or size + grads[j].numel() * grads[j].element_size() <= bucket_bytes):
            size += grads[j].numel() * grads[j].element_size()
            j += 1
        bucket = grads[i:j]
        flat = torch.cat([g.reshape(-1) for g in bucket])
        comm = flat if dist.get_backend() == "nccl" else flat.cpu()
        dist.all_reduce(comm, op=dist.ReduceOp.SUM)
        comm = comm.to(flat.device) / ws
        off = 0
        for g in bucket:
            g.copy_(comm[off:off + g.numel()].view_as(g))
            off += g.numel()
        calls += 1
        i = j
    return calls
