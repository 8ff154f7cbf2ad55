"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The reference is single-process ``nn.DataParallel`` (/root/reference/tools/train_3d.py:140):
scatter batch -> broadcast all parameters -> threads -> gather -> reduce-add grads, all through
GPU 0.  Here frames are sharded by rank (the path has no cross-frame dependency, SURVEY.md
§8(e)): inference needs NO collective; training all-reduces gradients only (DDP buckets over
RCCL/xGMI, overlapped with backward).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def env_world():
    """(rank, world, local_rank) from torchrun's environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str | None = None, device: torch.device | None = None):
    rank, world, _ = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def shard_frames(num_frames: int, rank: int, world: int) -> List[int]:
    """DistributedSampler semantics without padding: rank r owns frames r, r+world, ..."""
    return list(range(rank, num_frames, world))


def shard_batch(tensors: Sequence[torch.Tensor], rank: int, world: int):
    """slice dim 0 of every tensor the way ``shard_frames`` assigns frames"""
    return [t[rank::world] for t in tensors]


def max_over_ranks(value: float, device=None) -> float:
    """bench timing rule: the job takes as long as its slowest rank"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_predictions(pred: torch.Tensor) -> torch.Tensor:
    """all-gather per-rank predictions (B_r, ...) back into frame order (evaluation only)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pred
    world = dist.get_world_size()
    parts = [torch.empty_like(pred) for _ in range(world)]
    dist.all_gather(parts, pred.contiguous())
    out = torch.stack(parts, dim=1)                 # (B_r, world, ...): frame f = i*world + r
    return out.reshape(-1, *pred.shape[1:])


def wrap_ddp(model: torch.nn.Module, device: torch.device | None = None, find_unused: bool = True):
    """DistributedDataParallel with gradient-only all-reduce; no-op for a single process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    if device is not None and device.type == "cuda":
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], output_device=device.index,
                                                         find_unused_parameters=find_unused, bucket_cap_mb=32,
                                                         gradient_as_bucket_view=True)
    return torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=find_unused)
