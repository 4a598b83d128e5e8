"""Multi-GPU data parallelism over temporal batches (one process per GPU, torch.distributed = RCCL over xGMI).

The reference's only live multi-GPU mode is process-level data parallelism in the CLI: one mp.Process
per GPU, results returned through mp.Queue as shared-memory CPU tensors (inference_cli.py:1127-1288).
Temporal batches are independent through encode -> DiT -> decode, so the native form is:
batch i -> rank i mod world (keeps batch boundaries, hence pixels, identical to the single-GPU run),
weights replicated, no data-path collective until ONE all-gather of the upscaled bf16 THWC frames (plus, with
temporal overlap, one point-to-point message per batch boundary between the two owners).
xGMI is fully connected point-to-point, so the all-gather is issued as a single collective per step
(each peer link carries only that peer's shard).
"""
import os
from typing import List, Optional, Sequence

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


def all_gather_frames(local: torch.Tensor, force: bool = False) -> torch.Tensor:
    """local [F, H, W, 3] (same F on every rank) -> [world * F, H, W, 3] in rank order."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out


def _agree_on_like(owned: Sequence[torch.Tensor], device) -> tuple:
    """(shape, dtype) of one batch output, also on ranks that own none: a tiny MAX all-reduce of [dtype code, ndim, dims...]."""
    codes = [torch.bfloat16, torch.float32, torch.float16, torch.float64]
    info = torch.zeros(10, dtype=torch.int64, device=device)
    if owned:
        like = owned[0]
        info[0] = codes.index(like.dtype) + 1
        info[1] = like.dim()
        info[2:2 + like.dim()] = torch.tensor(like.shape, dtype=torch.int64)
    dist.all_reduce(info, op=dist.ReduceOp.MAX)
    vals = info.tolist()
    if vals[0] == 0:
        raise ValueError("gather_batches: no rank owns a batch")
    return tuple(vals[2:2 + vals[1]]), codes[vals[0] - 1]


def gather_batches(owned: Sequence[torch.Tensor], owned_idx: Sequence[int], n_batches: int,
                   device: Optional[torch.device] = None) -> List[torch.Tensor]:
    """Reassemble per-batch outputs of equal shape from all ranks in batch order.  Ranks with fewer batches -- or none,
    when n_batches < world -- contribute a zero placeholder that is dropped (``device`` tells such a rank where)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(owned)
    world = dist.get_world_size()
    per_rank = (n_batches + world - 1) // world
    dev = owned[0].device if owned else torch.device(device if device is not None else
                                                     ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    shape, dtype = _agree_on_like(owned, dev)
    F = shape[0]
    result = [None] * n_batches
    for j in range(per_rank):
        mine = owned[j] if j < len(owned) else torch.zeros(shape, dtype=dtype, device=dev)
        gathered = all_gather_frames(mine)
        for r in range(world):
            b = j * world + r
            if b < n_batches:
                result[b] = gathered[r * F:(r + 1) * F]
    return result


_INT_VIEW = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}     # same-width integer view (gloo has no bf16 transport)


def exchange_heads_p2p(heads: dict, boundaries: Sequence[int], shape: tuple, device, dtype=torch.bfloat16) -> dict:
    """Overlap heads travel point to point: the owner of batch i (rank i mod world) sends the first ``overlap`` decoded
    frames of batch i to the owner of batch i-1, which blends them into its tail (pipeline.upscale phase 3).  One
    message of overlap x H x W x 3 in the pipeline's storage dtype (``dtype``: bf16 on the HIP path -- 149 MB per boundary at
    4K / overlap 3 --, fp32 over the fp32 double of the C ABI, so the sharded run stays bit-identical to the single-rank run
    in every regime) per batch boundary over the direct xGMI link of that rank pair, instead of a dense all-reduce over every
    batch of the clip (round 1: 2.4 GB at BASELINE config 4, growing with clip length).  Returns {i: head} for the heads THIS
    rank has to blend."""
    rank, world = dist.get_rank(), dist.get_world_size()
    iv = _INT_VIEW[torch.empty((), dtype=dtype).element_size()]
    ops, recv = [], {}
    for i in boundaries:                               # every rank walks the same boundary list in the same order
        src, dst = i % world, (i - 1) % world
        if src == dst:
            continue
        if rank == src:
            ops.append(dist.P2POp(dist.isend, heads[i].to(device=device, dtype=dtype).contiguous().view(iv), dst))
        elif rank == dst:
            buf = torch.empty(shape, dtype=dtype, device=device)
            recv[i] = buf
            ops.append(dist.P2POp(dist.irecv, buf.view(iv), src))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return recv


def gather_to_root(padded: torch.Tensor, root: int = 0) -> Optional[torch.Tensor]:
    """[cap, ...] from every rank -> [world, cap, ...] on ``root`` only (None elsewhere): world - 1 point-to-point messages
    into the root instead of an all-gather that delivers every share to every rank -- 1/world of the traffic when only one
    process consumes the clip (the CLI's rank 0 writes the video; inference_cli.py:1166-1288 returns results to the parent
    process the same way)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    iv = _INT_VIEW[padded.element_size()]
    padded = padded.contiguous()
    if rank != root:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, padded.view(iv), root)]):
            req.wait()
        return None
    out = torch.empty((world,) + tuple(padded.shape), dtype=padded.dtype, device=padded.device)
    out[root] = padded
    ops = [dist.P2POp(dist.irecv, out[r].view(iv), r) for r in range(world) if r != root]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def upscale_sharded(images_thwc: torch.Tensor, runner, text_pos: torch.Tensor, **kw) -> Optional[torch.Tensor]:
    """pipeline.upscale() with the temporal batches dealt round-robin to the ranks of the default process group;
    every rank returns the complete clip, identical to the single-rank result (same batch boundaries, the overlap
    blend done by the owner of the blended frames).  Communication: one point-to-point message per batch boundary
    (only with temporal_overlap > 0, exchange_heads_p2p) and ONE all-gather of the upscaled bf16 frames.
    ``gather="root"``: only rank 0 receives the clip (the others return None) -- gather_to_root instead of the all-gather."""
    from . import pipeline
    force = kw.pop("force_collectives", False)         # tests: walk the collective code path even in a one-rank group
    gather = kw.pop("gather", "all")
    if gather not in ("all", "root"):
        raise ValueError("gather must be 'all' or 'root'")
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return pipeline.upscale(images_thwc, runner, text_pos, **kw)
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = runner.dit.device

    def exchange(heads: dict, boundaries: list, shape: tuple, dtype=torch.bfloat16) -> dict:
        return exchange_heads_p2p(heads, boundaries, shape, dev, dtype)

    final, spans = pipeline.upscale(images_thwc, runner, text_pos, batch_filter=lambda i: i % world == rank,
                                    exchange_heads=exchange, return_spans=True, **kw)
    # gather the frames each rank produced (padded to the largest share), then place them by span
    mine = torch.cat([final[a:b] for _, (a, b) in sorted(spans.items())], dim=0) if spans else final[:0]
    counts = torch.zeros(world, dtype=torch.int64, device=final.device)
    counts[rank] = mine.shape[0]
    dist.all_reduce(counts)
    cap = int(counts.max())
    padded = torch.zeros((cap,) + tuple(final.shape[1:]), dtype=final.dtype, device=final.device)
    padded[:mine.shape[0]] = mine
    table = torch.full((final.shape[0],), -1, dtype=torch.int64, device=final.device)          # frame -> owner rank
    for _, (a, b) in spans.items():
        table[a:b] = rank
    owner = table.clone()
    dist.all_reduce(owner, op=dist.ReduceOp.MAX)
    if gather == "root":
        gathered = gather_to_root(padded)
        if gathered is None:
            return None
    else:
        gathered = all_gather_frames(padded, force).reshape((world, cap) + tuple(final.shape[1:]))
    out = torch.empty_like(final)
    for r in range(world):
        idx = (owner == r).nonzero(as_tuple=True)[0]             # ascending = the order rank r concatenated them in
        out[idx] = gathered[r, :idx.numel()]
    return out
