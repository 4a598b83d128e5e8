"""-m "not gpu": the N>1 path (batch sharding + all-gather of frames) with 2 gloo processes on CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import sub, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_batches, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from conftest import sub as _sub
    d = _sub("dist")
    r, w, _ = d.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    owned_idx = d.batch_indices(n_batches, r, w)
    # "upscaled frames" of batch b: a [2, 3, 4, 3] tensor filled with b (bf16 like the product)
    owned = [torch.full((2, 3, 4, 3), float(b), dtype=torch.bfloat16) for b in owned_idx]
    full = d.gather_batches(owned, owned_idx, n_batches)
    ok = all(float(full[b].float().mean()) == float(b) for b in range(n_batches))
    g = d.all_gather_frames(torch.full((2, 3, 4, 3), float(rank), dtype=torch.bfloat16))
    ok = ok and g.shape == (2 * world, 3, 4, 3) and float(g[2 * rank].float().mean()) == rank
    q.put((rank, ok, owned_idx))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_batches", [4, 5])
def test_two_rank_batch_sharding_and_allgather(n_batches):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_batches, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    owned = sorted(sum((idx for _, _, idx in res), []))
    assert owned == list(range(n_batches))            # every batch owned exactly once


def test_split_frames_and_round_robin():
    d = sub("dist")
    assert d.split_frames(128, 17) == [(i * 17, min((i + 1) * 17, 128)) for i in range(8)]
    assert d.batch_indices(8, 3, 8) == [3] and d.batch_indices(5, 1, 2) == [1, 3]


def _empty_owner_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from conftest import sub as _sub
    d = _sub("dist")
    d.init_from_env(backend="gloo")
    owned_idx = d.batch_indices(1, rank, world)                       # rank 1 owns nothing
    owned = [torch.full((2, 3, 4, 3), 5.0, dtype=torch.bfloat16) for _ in owned_idx]
    full = d.gather_batches(owned, owned_idx, 1, device="cpu")
    q.put((rank, len(full) == 1 and float(full[0].float().mean()) == 5.0 and full[0].shape == (2, 3, 4, 3)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_batches_with_a_rank_that_owns_nothing():
    """n_batches < world: the idle rank learns shape and dtype from the group and contributes a placeholder (round 1
    indexed owned[0] there and deadlocked the others in the all-gather)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_owner_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _pipeline_worker(rank, world, port, q, noise=0.0, case=None, gather="all"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from conftest import sub as _sub
    from test_glue import _IdentityRunner
    d = _sub("dist")
    d.init_from_env(backend="gloo")
    frames, kw = case or (23, dict(resolution=32, batch_size=7, uniform_batch_size=True, temporal_overlap=2, color_correction="wavelet"))
    images = torch.rand(frames, 16, 24, 3, generator=torch.Generator().manual_seed(5))
    out = d.upscale_sharded(images, _IdentityRunner(), torch.zeros(58, 8), input_noise_scale=noise, gather=gather, **kw)
    # (by value: a tensor in an mp.Queue travels as a shared-memory handle the parent must fetch while this process is still alive --
    # under load the worker could exit first and the parent's q.get() died with FileNotFoundError)
    q.put((rank, None if out is None else out.float().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,noise", [(2, 0.0), (3, 0.0), (2, 0.5)])
def test_sharded_pipeline_equals_single_rank_with_overlap_blend(world, noise):
    """Sharded four-phase pipeline (batches dealt round-robin, overlap heads sent point to point to the neighbouring
    batch's owner, frames all-gathered) == the single-rank pipeline, bit for bit, on every rank; with input noise the
    ranks draw-and-discard the noise of the batches they skip, so every batch sees the single-rank run's noise."""
    from test_glue import _IdentityRunner
    pipeline = sub("pipeline")
    images = torch.rand(23, 16, 24, 3, generator=torch.Generator().manual_seed(5))
    want = pipeline.upscale(images, _IdentityRunner(), torch.zeros(58, 8), resolution=32, batch_size=7,
                            uniform_batch_size=True, temporal_overlap=2, color_correction="wavelet",
                            input_noise_scale=noise).float()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q, noise)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, out in res:
        assert out.shape == want.shape and torch.equal(torch.from_numpy(out), want)


CFG4_PLAN = (128, dict(resolution=32, batch_size=17, uniform_batch_size=True, temporal_overlap=1, color_correction="lab"))


@pytest.mark.parametrize("gather", ["all", "root"])
def test_world8_cfg4_plan_equals_single_rank(gather):
    """BASELINE config 4's plan on 8 ranks (what bench.py --workload cfg4 --gpus 8 runs): a 128-frame clip = 8 temporal batches
    of 17 with a 1-frame overlap (the last one 9 frames, padded to 17), one batch per rank, 7 point-to-point overlap heads
    between neighbouring ranks, then the frame gather -- to every rank, or (``gather="root"``, the CLI's case) to rank 0 only.
    Bit-equal to the single-rank pipeline."""
    from test_glue import _IdentityRunner
    pipeline = sub("pipeline")
    frames, kw = CFG4_PLAN
    plans, ov = pipeline.plan_batches(frames, kw["batch_size"], kw["temporal_overlap"], kw["uniform_batch_size"])
    assert len(plans) == 8 and ov == 1 and plans[-1].end - plans[-1].start == 16 and plans[-1].uniform_pad == 1
    images = torch.rand(frames, 16, 24, 3, generator=torch.Generator().manual_seed(5))
    want = pipeline.upscale(images, _IdentityRunner(), torch.zeros(58, 8), **kw).float()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 8, port, q, 0.0, CFG4_PLAN, gather)) for r in range(8)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, out in res.items():
        if gather == "root" and r != 0:
            assert out is None
        else:
            assert out.shape == want.shape and torch.equal(torch.from_numpy(out), want), r
