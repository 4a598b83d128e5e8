"""-m gpu: parity AT THE HEADLINE SCALE (BASELINE config 3: 33 frames 2160x3840, N = 9 x 135 x 240 = 291 600 video tokens, 33-frame
1024^2 VAE tiles) -- the sizes bench.py times, where the tensors pass 2^31 elements / 2^32 bytes and the other fixtures (crops of at
most 9 x 30 x 54 tokens, 5 x 1024 x 1152 px) cannot see a 32-bit index slip.

  (i)   NaDiT blocks at production width over ALL 291 600 tokens: an MM block with regular windows, a shared-weight block with
        shifted (ragged) windows and the final video-only block run through ``NaDiTEngine.forward`` twice -- over HipOps and over the
        fp32 torch restatement of the C ABI (tests/ops_reference.TorchOps on the same GPU, same packed weights, same storage kinds)
        -- and the residual stream behind every block is compared over EVERY row: globally, per attention window of that block (a
        slip confined to one window cannot hide in the average) and per row.  The qkv tensor of this run holds 2.24e9 elements.
        Reference path: src/models/dit_3b/nablocks/attention/mmattn.py:161-271, mmsr_block.py:108-126, na.py:320-424.
  (ii)  The decoder's full-resolution level on a 33 x 1024 x 1024 tile (8.9 GB per 128-channel tensor, 17.7 GB at 256): both kinds of
        ResnetBlock3D (256 -> 128 with its 1x1x1 shortcut -- a plain GEMM over 34.6 M rows --, 128 -> 128), conv_norm_out and conv_out
        run through the engine's own layer code over HipOps; EVERY conv / GEMM launch and GroupNorm pass it issued is then re-computed in fp32 (F.conv3d / the GroupNorm
        formula) on crops of the launch's own input with halo -- the four corners of the first and of the last frame, the last rows
        and columns, crops behind byte offset 2^32 and element offset 2^32 -- and every set of fused GroupNorm statistics against an
        fp64 reduction of the full stored tensor.  Reference path: causal_inflation_lib.py:213-305, 354-409, attn_video_vae.py:255-362.
  (iii) Tile blending on the 4K canvas (32 x 2160 x 3840 x 3 fp32 = 3.2e9 bytes): a tile blended at the far corner of the canvas equals
        the same tile blended at the origin bit for bit, finalize matches torch, and the one-rank RCCL all-gather of the canvas returns
        it bit for bit.  Reference path: attn_video_vae.py:1594-1625, inference_cli.py:1127-1288.
Tolerances as everywhere (tests/test_gpu_kernels.py): <= 2.5e-3 for a bf16 store, <= 1e-3 for an fp32 / h16 store, <= 4e-3 behind attention.
"""
import copy
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import sub, rel_err
from ops_reference import TorchOps, _ld

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
TOL_BF16, TOL_WIDE, TOL_ATTN = 2.5e-3, 1e-3, 4e-3


@pytest.fixture(scope="module")
def hip():
    return sub("ops").HipOps("cuda:0")


def _free():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ (i) NaDiT blocks over all 291 600 tokens
def test_dit_blocks_at_headline_token_count_every_row(hip):
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    cfg = config.DiTConfig(num_layers=3, mm_layers=1)       # block 0: MM weights, regular windows; 1: shared weights, shifted
    dev = hip.device                                       # windows, both streams in one GEMM; 2: the final video-only block
    T, H, W = 9, 270, 480                                   # latent grid of 33 frames 2160 x 3840
    t, h, w = T, H // 2, W // 2
    N = t * h * w
    assert N == 291600
    inner = cfg.heads * cfg.head_dim
    assert (N + 58) * 3 * inner > 2 ** 31                   # qkv: element offsets beyond int32
    sd = weights.synth_dit_state_dict(cfg, device=dev)
    eng = dit.NaDiTEngine(cfg, sd, hip)
    g = torch.Generator(device=dev).manual_seed(7)
    vid = torch.randn(T, H, W, cfg.vid_in_channels, generator=g, device=dev).to(BF16)
    txt = weights.synth_text_embedding(device=dev)
    Lt = txt.shape[0]

    taps = {}
    eng.tap = lambda tag, x: taps.__setitem__(tag, x[:N + Lt].clone())
    got = eng.forward(vid, txt, 1000.0)
    torch.cuda.synchronize()
    got_taps, taps = taps, {}
    ref_eng = copy.copy(eng)                                # same packed weights and index plans, the torch restatement of every op
    ref_eng.ops = TorchOps(dev)
    ref_eng.tap = lambda tag, x: taps.__setitem__(tag, x[:N + Lt].clone())
    want = ref_eng.forward(vid, txt, 1000.0)
    torch.cuda.synchronize()

    def row_err(a, b):
        a, b = a.float(), b.float()
        return (a - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-20)

    for li in range(1, cfg.num_layers + 1):
        a, b = got_taps[f"hid{li}"], taps[f"hid{li}"]
        rows = N + Lt if li < cfg.num_layers else N         # (the text stream is dead behind the last block's attention)
        e_all = rel_err(a[:rows], b[:rows])
        e_row = row_err(a[:rows], b[:rows])
        # per attention window of the block that produced this state (window membership from the engine's own index plan)
        plan = eng._plan((t, h, w), cfg.window_method(li - 1), Lt)
        cu, seq = plan["cu"].tolist(), plan["seq_rows"].long()
        worst_w, worst = -1, 0.0
        for wi in range(plan["n_win"]):
            r = seq[cu[wi]:cu[wi + 1] - Lt]                 # the window's video rows
            e = rel_err(a[r], b[r])
            if e > worst:
                worst_w, worst = wi, e
        print(f"block {li - 1} ({cfg.window_method(li - 1)}, {plan['n_win']} windows, longest {plan['max_len']} rows): stream rel-err "
              f"{e_all:.3e}; worst window #{worst_w} {worst:.3e}; worst row {float(e_row.max()):.3e}; text rows "
              f"{rel_err(a[N:rows], b[N:rows]) if rows > N else float('nan'):.3e}")
        assert e_all < TOL_ATTN and worst < 1.5 * TOL_ATTN and float(e_row.max()) < 5e-2
        if rows > N:
            assert rel_err(a[N:rows], b[N:rows]) < TOL_ATTN
    e = rel_err(got.float(), want.float())
    e_tok = row_err(got.reshape(-1, got.shape[-1]), want.reshape(-1, want.shape[-1]))
    print(f"prediction [9, 270, 480, 16] after 3 blocks: rel-err {e:.3e}, worst voxel {float(e_tok.max()):.3e}")
    assert e < 6e-3 and torch.isfinite(got.float()).all()


# ------------------------------------------------------------------ (ii) the decoder's full-resolution level on a 33 x 1024^2 tile
def _recording_ops(hip_cls, dev):
    class RecordingOps(hip_cls):
        def __init__(self, device):
            super().__init__(device)
            self.convs, self.norms, self.gemms = [], [], []

        def gemm(self, A, W, out, **kw):
            r = super().gemm(A, W, out, **kw)
            if kw.get("conv") is not None:
                self.convs.append(dict(A=A, W=W, out=out, kw=kw, stats=r[1] if isinstance(r, tuple) else None))
            else:
                self.gemms.append(dict(A=A, W=W, out=out, kw=kw))
            return r

        def groupnorm_apply(self, x, out, stats, gamma, beta, groups, eps, silu):
            self.norms.append(dict(x=x, out=out, stats=stats, gamma=gamma, beta=beta, groups=groups, eps=eps, silu=silu))
            return super().groupnorm_apply(x, out, stats, gamma, beta, groups, eps, silu)

    return RecordingOps(dev)


def _crops(To, Ho, Wo):
    """Output crops (t0, t1, y0, y1, x0, x1): corners of the first and last frames (the tile borders: zero padding, ragged 16 x 32
    patches), the last rows / columns, an interior crop straddling patch and band boundaries, and crops in the frames whose byte /
    element offsets pass 2^32 (frame 16 of a 128-channel bf16 tensor starts at byte 2^32, frame 32 at element 2^32)."""
    ch, cw = 20, 40
    tl = To - 1
    mid_t = min(16, tl)
    out = [(0, min(2, To), 0, ch, 0, cw), (0, min(2, To), Ho - ch, Ho, Wo - cw, Wo),
           (tl, To, 0, ch, Wo - cw, Wo), (tl, To, Ho - ch, Ho, 0, cw), (tl, To, Ho - ch, Ho, Wo - cw, Wo),
           (max(tl - 1, 0), To, Ho // 2 - 7, Ho // 2 + 13, Wo // 2 - 11, Wo // 2 + 29),
           (mid_t, min(mid_t + 2, To), Ho - 3, Ho, 0, Wo // 4), (mid_t, min(mid_t + 1, To), 0, Ho // 8, Wo - 3, Wo),
           (To // 2, To // 2 + 1, 500 % Ho, 500 % Ho + ch, 700 % Wo, min(700 % Wo + cw, Wo))]
    return [c for c in out if c[1] > c[0] and c[3] > c[2] and c[5] > c[4]]


def _conv_crop_reference(rec, crop):
    """fp32 F.conv3d of one recorded launch on an output crop, from the launch's own input with halo (TorchOps.gemm's semantics:
    causal head = the carried halo frames or replicated frame 0, spatial zero padding, bias, residual)."""
    kw, A = rec["kw"], rec["A"]
    g, N = kw["conv"], kw["N"]
    t0, t1, y0, y1, x0, x1 = crop
    kt, kh, kw_ = g.k
    st, sh, sw = g.stride
    pt, ph, pw = g.pad
    x = A.reshape(g.T, g.H, g.W, g.Cin)
    frames = []
    for p in range(t0 * st, (t1 - 1) * st + kt):           # padded-time index -> input frame
        f = p - pt
        if f >= 0:
            frames.append(x[f])
        elif g.halo is not None:
            frames.append(g.halo[g.halo.shape[0] + f])
        else:
            frames.append(x[0])
    ya, yb = y0 * sh - ph, (y1 - 1) * sh - ph + kh          # input rows [ya, yb), columns [xa, xb); zero outside the image
    xa, xb = x0 * sw - pw, (x1 - 1) * sw - pw + kw_
    ya_c, yb_c, xa_c, xb_c = max(ya, 0), min(yb, g.H), max(xa, 0), min(xb, g.W)
    blk = torch.stack([f[ya_c:yb_c, xa_c:xb_c] for f in frames]).float()               # [kt', h, w, Cin]
    blk = F.pad(blk.permute(3, 0, 1, 2), (xa_c - xa, xb - xb_c, ya_c - ya, yb - yb_c))  # [Cin, kt', h', w']
    K = kt * kh * kw_ * g.Cin
    w5 = rec["W"][:N, :K].float().reshape(N, kt, kh, kw_, g.Cin).permute(0, 4, 1, 2, 3)
    y = F.conv3d(blk[None], w5, stride=(st, sh, sw))[0].permute(1, 2, 3, 0)              # [t, y, x, N]
    assert tuple(y.shape[:3]) == (t1 - t0, y1 - y0, x1 - x0), (y.shape, crop)
    if kw.get("bias") is not None:
        y = y + kw["bias"][:N].float()
    if kw.get("resid") is not None:
        y = y + _ld(kw["resid"].reshape(g.To, g.Ho, g.Wo, -1)[t0:t1, y0:y1, x0:x1, :N])
    return y


def test_decoder_full_resolution_level_on_a_33_frame_1024px_tile(hip):
    config, weights, vae_mod, opsmod = sub("config"), sub("weights"), sub("vae"), sub("ops")
    dev = hip.device
    vcfg = config.VAE_V3
    rec = _recording_ops(opsmod.HipOps, dev)
    eng = vae_mod.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, device=dev), rec)
    T, S = 33, 1024
    res, up = eng.dec_up[len(eng.dec_up) - 1]
    assert up is None and res[0].shortcut is not None and res[1].shortcut is None
    cin = res[0].conv1.cin                                  # 256 channels in, 128 out
    assert T * S * S * res[1].conv1.cin * 2 > 2 ** 32 and T * S * S * res[1].conv1.cin > 2 ** 32
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.empty(T, S, S, cin, dtype=BF16, device=dev)   # block input as the upsampler leaves it: bf16 (a shortcut conv reads it)
    for f in range(T):                                      # (frame by frame: no 70 GB fp32 temporary)
        x[f] = (torch.randn(S, S, cin, generator=g, device=dev) * (1.0 + 0.25 * math.sin(f)) + 0.1 * f / T).to(BF16)
    st = {"__last_slice__": True}
    rec.convs.clear(); rec.norms.clear()
    h, hs = eng._resnet(res[0], x, st, True, None, wide=True)          # up_blocks.3.resnets.0 (shortcut conv) -> trunk (h16)
    h, hs = eng._resnet(res[1], h, st, True, hs, wide=True)            # up_blocks.3.resnets.1
    h = eng._gn(eng.dec_norm_out, h, True, hs)
    y = eng._conv(eng.dec_conv_out, h, st, True)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (T, S, S, 3) and bool(torch.isfinite(y.float()).all())
    kinds = set()
    worst = {}
    for r in rec.convs:
        kw = r["kw"]
        gm = kw["conv"]
        out = r["out"].reshape(gm.To, gm.Ho, gm.Wo, -1)
        tol = TOL_BF16 if out.dtype == BF16 else TOL_WIDE
        name = f"conv {gm.Cin}->{kw['N']} k{gm.k} T{gm.T}->{gm.To} {str(out.dtype).split('.')[-1]}" + (" +resid" if kw.get("resid") is not None else "")
        kinds.add(name)
        for crop in _crops(gm.To, gm.Ho, gm.Wo):
            t0, t1, y0, y1, x0, x1 = crop
            want = _conv_crop_reference(r, crop)
            e = rel_err(_ld(out[t0:t1, y0:y1, x0:x1, :kw["N"]]), want)
            worst[name] = max(worst.get(name, 0.0), e)
            assert e < tol, (name, crop, e)
        if r["stats"] is not None:                          # fused GroupNorm statistics == fp64 sums of the tensor the launch stored
            cpg = kw["N"] // 32
            for f in range(gm.To):
                v = _ld(out[f]).double().reshape(-1, 32, cpg)
                want = torch.stack([v.sum(dim=(0, 2)), (v * v).sum(dim=(0, 2))], dim=-1)
                assert torch.allclose(r["stats"][f], want, rtol=2e-6, atol=1e-3), (name, f)
    for name, e in sorted(worst.items()):
        print(f"{name}: worst crop rel-err {e:.3e}")
    # (conv1 / conv2 of both blocks and conv_out as a causal-head + body launch pair; the 1x1x1 shortcut is a plain GEMM over the rows)
    assert len(rec.convs) == 2 * 2 + 2 * 2 + 2 and len(kinds) >= 4 and len(rec.gemms) == 1
    for r in rec.gemms:                                     # the shortcut: [34.6 M rows, 256] @ [256, 128]^T + bias -> h16, row blocks up to the last row
        kw = r["kw"]
        M_, N_, K_ = kw["M"], kw["N"], kw["K"]
        A2, O2 = r["A"].reshape(M_, -1), r["out"].reshape(M_, -1)
        assert M_ == T * S * S and M_ * K_ * 2 > 2 ** 32
        for a0 in (0, 12345 * 7, M_ // 2 + 3, (2 ** 32) // (K_ * 2) - 100, (2 ** 32) // N_ + 1000, M_ - 5000):
            a1 = min(a0 + 5000, M_)
            want = A2[a0:a1, :K_].float() @ r["W"][:N_, :K_].float().t() + kw["bias"][:N_].float()
            e = rel_err(_ld(O2[a0:a1, :N_]), want)
            assert e < TOL_WIDE, ("shortcut GEMM", a0, e)
        print(f"1x1x1 shortcut as a plain GEMM [{M_}, {K_}] x [{N_}, {K_}]^T -> {str(O2.dtype).split('.')[-1]}: row blocks up to the last row <= {TOL_WIDE}")
    for n_, r in enumerate(rec.norms):
        xs, out = r["x"], r["out"]
        Tn, Hn, Wn, Cn = xs.shape
        cpg = Cn // r["groups"]
        n_el = Hn * Wn * cpg
        mean = r["stats"][..., 0] / n_el
        rstd = 1.0 / torch.sqrt((r["stats"][..., 1] / n_el - mean * mean).clamp_min(0) + r["eps"])
        # the statistics themselves (fused or from svr_groupnorm_stats) against an fp64 reduction of two frames of the full tensor
        for f in (0, Tn - 1):
            v = _ld(xs[f]).double().reshape(-1, r["groups"], cpg)
            assert torch.allclose(r["stats"][f, :, 0], v.sum(dim=(0, 2)), rtol=2e-6, atol=1e-3)
            assert torch.allclose(r["stats"][f, :, 1], (v * v).sum(dim=(0, 2)), rtol=2e-6, atol=1e-3)
        worst_n = 0.0
        for (t0, t1, y0, y1, x0, x1) in _crops(Tn, Hn, Wn):
            m = mean[t0:t1].repeat_interleave(cpg, dim=1).float()[:, None, None, :]
            s = rstd[t0:t1].repeat_interleave(cpg, dim=1).float()[:, None, None, :]
            want = (_ld(xs[t0:t1, y0:y1, x0:x1]) - m) * s * r["gamma"] + r["beta"]
            if r["silu"]:
                want = F.silu(want)
            e = rel_err(out[t0:t1, y0:y1, x0:x1].float(), want)
            worst_n = max(worst_n, e)
            assert e < TOL_BF16, (n_, e)
        print(f"GroupNorm-apply #{n_} [{Tn}, {Hn}, {Wn}, {Cn}] {str(xs.dtype).split('.')[-1]} silu={r['silu']}: worst crop rel-err {worst_n:.3e}")
    assert len(rec.norms) == 5
    del rec, eng, x, h, y
    _free()


# ------------------------------------------------------------------ (iii) tile blending and the all-gather on the 4K canvas
def test_blend_and_allgather_on_the_4k_canvas(hip):
    dev = hip.device
    T, H, W, Cc = 32, 2160, 3840, 3
    assert T * H * W * Cc * 4 > 2 ** 31
    th, tw = 1024, 1024
    g = torch.Generator(device=dev).manual_seed(3)
    tile = torch.randn(T, th, tw, Cc, generator=g, device=dev).to(BF16)
    wy, wx = torch.rand(th, generator=g, device=dev), torch.rand(tw, generator=g, device=dev)
    acc, cnt = torch.zeros(T, H, W, Cc, device=dev), torch.zeros(H, W, device=dev)
    y0, x0 = H - th, W - tw                                 # the far corner: the largest offsets of the canvas
    hip.blend_accumulate(tile, acc, cnt, wy, wx, 0, 0)
    hip.blend_accumulate(tile, acc, cnt, wy, wx, y0, x0)
    torch.cuda.synchronize()
    assert torch.equal(acc[:, :th, :tw], acc[:, y0:, x0:]) and torch.equal(cnt[:th, :tw], cnt[y0:, x0:])
    wgt = wy[:, None] * wx[None, :]
    assert torch.allclose(acc[:, y0:, x0:], tile.float() * wgt[None, :, :, None], rtol=1e-6, atol=1e-7)
    assert float(acc[:, th:y0].abs().max()) == 0.0 and float(acc[:, :, tw:x0].abs().max()) == 0.0   # nothing written elsewhere
    cnt += (cnt == 0).float()                               # (uncovered pixels: finalize divides by the count)
    out = torch.empty(T, H, W, Cc, dtype=BF16, device=dev)
    hip.blend_finalize(acc, cnt, out)
    want = TorchOps(dev).blend_finalize(acc, cnt, torch.empty(T, H, W, Cc, device=dev))
    assert rel_err(out[:, y0:, x0:].float(), want[:, y0:, x0:]) < TOL_BF16 and rel_err(out[:, :th, :tw].float(), want[:, :th, :tw]) < TOL_BF16
    del want
    # the gather of the upscaled frames, one rank: RCCL returns the canvas bit for bit (world 2 is test_dist's, on 2-GPU boxes)
    import torch.distributed as dist
    dist_mod = sub("dist")
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29917")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        gathered = dist_mod.all_gather_frames(acc, force=True)
        torch.cuda.synchronize()
        assert gathered.shape[0] == T and torch.equal(gathered.reshape(acc.shape), acc)
    finally:
        if own:
            dist.destroy_process_group()
    del acc, out, gathered
    _free()
