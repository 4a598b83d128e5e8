"""Multi-GPU data parallelism over temporal batches (one process per GPU, torch.distributed = RCCL over xGMI).

The reference's only live multi-GPU mode is process-level data parallelism in the CLI: one mp.Process
per GPU, results returned through mp.Queue as shared-memory CPU tensors (inference_cli.py:1127-1288).
Temporal batches are independent through encode -> DiT -> decode, so the native form is:
batch i -> rank i mod world (keeps batch boundaries, hence pixels, identical to the single-GPU run),
weights replicated, no data-path collective until ONE all-gather of the upscaled bf16 THWC frames.
xGMI is fully connected point-to-point, so the all-gather is issued as a single collective per step
(each peer link carries only that peer's shard).
"""
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> tuple:
    """(rank, world, local_rank).  Initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def batch_indices(n_batches: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership of temporal batches."""
    return list(range(rank, n_batches, world))


def split_frames(n_frames: int, batch_size: int) -> List[tuple]:
    """[start, stop) frame ranges of the temporal batches (last one may be shorter)."""
    return [(s, min(s + batch_size, n_frames)) for s in range(0, n_frames, batch_size)]


def all_gather_frames(local: torch.Tensor) -> torch.Tensor:
    """local [F, H, W, 3] (same F on every rank) -> [world * F, H, W, 3] in rank order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out


def gather_batches(owned: Sequence[torch.Tensor], owned_idx: Sequence[int], n_batches: int) -> List[torch.Tensor]:
    """Reassemble per-batch outputs of equal shape from all ranks in batch order (ranks with fewer
    batches contribute a zero placeholder that is dropped)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(owned)
    world, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (n_batches + world - 1) // world
    like = owned[0]
    result = [None] * n_batches
    for j in range(per_rank):
        mine = owned[j] if j < len(owned) else torch.zeros_like(like)
        gathered = all_gather_frames(mine)
        F = like.shape[0]
        for r in range(world):
            b = j * world + r
            if b < n_batches:
                result[b] = gathered[r * F:(r + 1) * F]
    return result
