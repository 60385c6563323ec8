"""Multi-GPU sharding of the per-video path: one process per GPU, videos sharded
``videos[rank::world]`` (what ``DistributedSampler(shuffle=False)`` does in the reference,
``trainer_ddp.py:144``), ONE collective at the end: an all-gather of the per-rank score vectors
(``trainer_ddp.py:259-267``) — RCCL over xGMI on the GPU box (backend "nccl"), gloo in CPU tests.
No data-path collective exists: a video's score depends on no other video.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process -> 0,0,1)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def default_master_port() -> int:
    """Rendezvous port when the launcher exported none (torch.distributed.run always does): derived from the PARENT pid,
    which all ranks of one job share and two jobs on one node do not — a fixed 29500 makes concurrent jobs collide."""
    return 20000 + (os.getppid() * 7919) % 20000


def init(backend: str = None, force: bool = False) -> Tuple[int, int, int]:
    """Join the job's process group (world > 1, or ``force`` for a 1-rank group: lets the RCCL path run on a 1-GPU box)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(default_master_port()))
        if backend is None:
            backend = os.environ.get("KVQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Indices this rank scores, padded by wrap-around to ceil(n/world) like DistributedSampler
    (the padded tail is dropped again by ``gather_scores``)."""
    per = -(-n_items // world)
    return [(rank + i * world) % n_items for i in range(per)]


def gather_scores(local: torch.Tensor, n_items: int, rank: int, world: int) -> torch.Tensor:
    """local: fp32 [ceil(n/world)] scores of ``shard_indices`` -> fp32 [n_items] in item order on every
    rank.  One all_gather_into_tensor of a <1 KB vector: latency-bound, ring vs direct irrelevant."""
    per = -(-n_items // world)
    assert local.numel() == per
    if world == 1 and not dist.is_initialized():
        return local[:n_items].clone()
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:
        # gloo moves host memory only.  Device scores through it are a TEST arrangement (N ranks sharing one GPU): it has to
        # be asked for by name, a production job on GPUs runs RCCL ("nccl") and never stages through the host.
        if os.environ.get("KVQ_DIST_BACKEND") != "gloo":
            raise RuntimeError("score all-gather: the process group is gloo but the scores live on a GPU; use the nccl "
                               "backend (RCCL), or set KVQ_DIST_BACKEND=gloo to stage through the host on purpose")
        local = local.cpu()
    out = torch.empty(world * per, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    out = out.to(dev)
    # out[r*per + i] is item (r + i*world) % n; undo, keeping the first occurrence of every item
    full = torch.empty(n_items, dtype=out.dtype, device=dev)
    src = torch.arange(world * per, device=dev)
    item = (src // per + (src % per) * world)
    keep = item < n_items
    full[item[keep]] = out[keep]
    return full


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
