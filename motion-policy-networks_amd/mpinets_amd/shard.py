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
    """Join the job torchrun started and make this rank's GPU the current device -- for EVERY backend: the engine
    enqueues on torch's current stream of the current device, so a gloo job that skipped ``set_device`` would put every
    rank's kernels on device 0.  Backend: RCCL ("nccl") when GPUs are present, gloo otherwise; ``MPX_DIST_BACKEND``
    overrides.

    ``device`` defaults to ``cuda:LOCAL_RANK``.  RCCL needs one device per rank: with fewer visible GPUs than local ranks
    an "nccl" job fails here with a clear message.  A gloo job may share GPUs (tests run 2 ranks on the box's one GPU):
    its default device is ``cuda:(LOCAL_RANK mod device_count)``.

    A single process (WORLD_SIZE 1) does not create a process group -- unless ``MPX_DIST_FORCE=1``: then the group is
    created anyway (world size 1), so that barrier / max_over_ranks / gather_to_rank0 run on a REAL communicator
    (tests/test_gpu_shard.py exercises the RCCL calls of ``bench.py --gpus N`` on a one-GPU box this way)."""
    rank, ws, local = world()
    backend = backend or os.environ.get("MPX_DIST_BACKEND")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if device is None:
            if local >= ndev and backend == "nccl":
                raise RuntimeError(f"shard.init: LOCAL_RANK {local} but only {ndev} GPU(s) visible -- RCCL needs one device "
                                   "per rank (share GPUs only with backend='gloo')")
            device = torch.device("cuda", local % ndev)
        else:
            device = torch.device(device)
        torch.cuda.set_device(device)
    force = os.environ.get("MPX_DIST_FORCE") == "1"
    if (ws > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # torchrun (and bench.py's own launcher) always export the port.  Without one, a single forced rank takes a
            # free port of its own (two such processes on one box never collide on a fixed default); several ranks
            # cannot each pick one, so that is an error rather than a guess.
            if ws > 1:
                raise RuntimeError("shard.init: WORLD_SIZE > 1 but MASTER_PORT is not set -- launch the ranks with "
                                   "torch.distributed.run (or `python bench.py --gpus N`), which agree on a port")
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=ws, **kwargs)
    return rank, ws, local


def backend_name() -> Optional[str]:
    """"nccl" (= RCCL on ROCm) / "gloo" of the live process group, None for a plain single process."""
    return dist.get_backend() if dist.is_initialized() else None


def gather_objects(obj) -> list:
    """Small picklable per-rank records (device name, per-rank timings) -> the list over ranks, on every rank."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


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
    if dist.get_backend() != "nccl":
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else device


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(t: torch.Tensor) -> Optional[torch.Tensor]:
    """Final host gather: per-rank results -> concatenated along dim 0 on rank 0 (None elsewhere).  The ranks' leading
    dimensions may differ (``split_even`` shares of a strong-scaling run): shorter shares are padded for the collective
    and trimmed afterwards."""
    if not dist.is_initialized():
        return t
    rank, ws = dist.get_rank(), dist.get_world_size()
    t = t.contiguous().to(_comm_device(t.device))
    sizes = gather_objects(int(t.size(0)))
    top = max(sizes)
    if t.size(0) < top:
        t = torch.cat((t, t.new_zeros((top - t.size(0),) + tuple(t.shape[1:]))), dim=0)
    bufs = [torch.empty_like(t) for _ in range(ws)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0) if rank == 0 else None


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
