"""Image-parallel sharding of independent query pairs over the GPUs of one node.

The reference has no distributed code (one process, one GPU: train_KITTI.py:3); every (grd, sat)
pair is independent in eval mode (per-sample softmax, BN running stats), so the path shards
embarrassingly: one process per GPU, weights replicated, a contiguous slice of the query range per
rank and ONE collective per step - an all_gather of the compact per-query result (argmax index,
probability, cos, sin, angle = 20 bytes/query) over RCCL/xGMI (backend "nccl" on ROCm).  The
CPU/gloo path exists for tests only.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from torchrun's environment (no-op for world size 1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("CCVPE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `n_items` queries for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local: torch.Tensor, n_items: int | None = None) -> torch.Tensor:
    """all_gather per-query rows [b_local, F] from every rank into [sum b, F] in rank order.

    Shards may be ragged by one row (shard_range); rows are padded to the largest shard for the
    collective and trimmed afterwards.  With world size 1 this is the identity.
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if n_items is None:
        sizes = [local.shape[0]] * world
    else:
        sizes = [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]
    width = max(sizes)
    pad = local
    if local.shape[0] < width:
        pad = torch.cat([local, local.new_zeros((width - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    if dist.get_backend() == "gloo" and pad.is_cuda:   # test rigs only: gloo gathers through host memory
        host = pad.cpu().contiguous()
        out_h = host.new_empty((world * width,) + tuple(host.shape[1:]))
        dist.all_gather_into_tensor(out_h, host)
        out = out_h.to(local.device)
    else:
        out = local.new_empty((world * width,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, pad.contiguous())
    if all(s == width for s in sizes):
        return out
    return torch.cat([out[r * width: r * width + sizes[r]] for r in range(world)], dim=0)


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device: torch.device | str = "cpu") -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
