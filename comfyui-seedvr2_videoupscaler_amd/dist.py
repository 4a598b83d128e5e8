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


def upscale_sharded(images_thwc: torch.Tensor, runner, text_pos: torch.Tensor, **kw) -> torch.Tensor:
    """pipeline.upscale() with the temporal batches dealt round-robin to the ranks of the default process group;
    every rank returns the complete clip, identical to the single-rank result (same batch boundaries, the overlap
    blend done by the owner of the blended frames).  Collectives: one small all-reduce of the overlap heads (only
    with temporal_overlap > 0) and ONE all-gather of the upscaled bf16 frames."""
    from . import pipeline
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return pipeline.upscale(images_thwc, runner, text_pos, **kw)
    rank, world = dist.get_rank(), dist.get_world_size()

    def exchange(heads: dict, n_batches: int, shape: tuple) -> dict:
        dense = torch.zeros((n_batches,) + tuple(shape), dtype=torch.float32, device=images_thwc.device)
        for i, h in heads.items():
            dense[i] = h.float()                                   # disjoint owners: the sum is an exact copy
        dist.all_reduce(dense)
        return {i: dense[i] for i in range(n_batches)}

    final, spans = pipeline.upscale(images_thwc, runner, text_pos, batch_filter=lambda i: i % world == rank,
                                    exchange_heads=exchange, return_spans=True, **kw)
    # all-gather the frames each rank produced (padded to the largest share), then place them by span
    mine = torch.cat([final[a:b] for _, (a, b) in sorted(spans.items())], dim=0) if spans else final[:0]
    counts = torch.zeros(world, dtype=torch.int64, device=final.device)
    counts[rank] = mine.shape[0]
    dist.all_reduce(counts)
    cap = int(counts.max())
    padded = torch.zeros((cap,) + tuple(final.shape[1:]), dtype=final.dtype, device=final.device)
    padded[:mine.shape[0]] = mine
    gathered = all_gather_frames(padded).reshape((world, cap) + tuple(final.shape[1:]))
    table = torch.full((final.shape[0],), -1, dtype=torch.int64, device=final.device)          # frame -> owner rank
    for _, (a, b) in spans.items():
        table[a:b] = rank
    owner = table.clone()
    dist.all_reduce(owner, op=dist.ReduceOp.MAX)
    out = torch.empty_like(final)
    for r in range(world):
        idx = (owner == r).nonzero(as_tuple=True)[0]             # ascending = the order rank r concatenated them in
        out[idx] = gathered[r, :idx.numel()]
    return out
