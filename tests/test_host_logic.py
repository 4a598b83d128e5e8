"""-m "not gpu": host logic of the engines (index plans, weight packing / interleaving, modulation
bookkeeping, temporal slicing + causal halos, spatial tiling + blending, runner API) verified on CPU by
injecting the torch restatement of the C-ABI ops (tests/ops_reference.py) and comparing with the
reference golden outputs / the oracle.  fp32 storage makes this exact to ~1e-6."""
import os

import pytest
import torch

from conftest import sub, rel_err, GOLDEN
from ops_reference import TorchOps
from oracle import dit_oracle, vae_oracle


@pytest.fixture(scope="module")
def vae_setup():
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg)
    # (the reference's two-step upsampler: with fp32 storage the engine then reproduces the goldens to ~1e-6; the sub-pixel
    # form rounds MERGED weights to bf16 and has its own tests below)
    return cfg, sd, vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=torch.float32), merge_upsamplers=False, merge_causal_head=False)


def _two_step_upsampler(x, w1, b1, w3, b3, rz):
    """The reference's Upsample3D in torch: upscale_conv, "b (x y z c) f h w -> b c (f z) (h x) (w y)", remove_head on a temporal
    upsampler, replicate-first causal head, zero spatial padding, 3x3x3 conv (attn_video_vae.py:110-174)."""
    F = torch.nn.functional
    T, H, W, C = x.shape
    y = (x @ w1.t() + b1).reshape(T, H, W, 2, 2, rz, C).permute(0, 5, 1, 3, 2, 4, 6).reshape(rz * T, 2 * H, 2 * W, C)
    if rz == 2:
        y = torch.cat([y[:1], y[2:]], 0)
    z = y.permute(3, 0, 1, 2)[None]
    z = F.pad(torch.cat([z[:, :, :1]] * 2 + [z], 2), (1, 1, 1, 1))
    return F.conv3d(z, w3, b3)[0].permute(1, 2, 3, 0)


@pytest.mark.parametrize("rz", [1, 2])
def test_subpixel_merge_matches_the_two_step_upsampler(rz):
    """upscale_conv -> pixel shuffle -> (remove_head) -> causal 3x3x3 conv == small convs over the low-resolution input with
    merged weights and border-aware bias, exactly (fp64), for any temporal slicing, incl. one-voxel images where both borders
    coincide and clips shorter than the head patterns."""
    F = torch.nn.functional
    subpixel = sub("subpixel")
    gen = torch.Generator().manual_seed(rz)
    C, Cout = 4, 5
    for T, H, W, slices in ((5, 3, 4, [5]), (5, 3, 4, [1, 1, 1, 1, 1]), (6, 2, 2, [2, 3, 1]), (1, 1, 1, [1]), (2, 1, 3, [1, 1]), (4, 5, 1, [3, 1])):
        rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
        x, w1, b1, w3, b3 = rnd(T, H, W, C), rnd(4 * rz * C, C), rnd(4 * rz * C), rnd(Cout, C, 3, 3, 3), rnd(Cout)
        want = _two_step_upsampler(x, w1, b1, w3, b3, rz)
        got = torch.full_like(want, float("nan"))
        sets = {}
        t0 = 0
        for n in slices:
            for t in range(t0, t0 + n):
                for i in subpixel.output_frames(t, rz):
                    sig = subpixel.signature(i, rz)
                    if sig not in sets:
                        sets[sig] = subpixel.merge_upsampler(w1, b1, w3, b3, rz, sig)
                    srcs, parts = sets[sig]
                    frames = torch.stack([x[max(t + s, 0)] for s in srcs], 0)            # (the clamp = the replicated head)
                    xin = frames.permute(3, 0, 1, 2)[None]
                    for py, px, wm, bias, bb in parts:
                        o = F.conv3d(F.pad(xin, (1 - px, px, 1 - py, py)), wm.double())[0].permute(1, 2, 3, 0)
                        b = bias.double().expand(1, H, W, Cout).clone()
                        rb, cb = (H - 1 if py else 0), (W - 1 if px else 0)
                        b[:, rb] = bb[0].double()
                        b[:, :, cb] = bb[1].double()
                        b[:, rb, cb] = bb[2].double()
                        got[i:i + 1, py::2, px::2] = o + b
            t0 += n
        assert not torch.isnan(got).any()
        assert rel_err(got, want) < 2e-6                        # (the merge itself works in fp32)
        assert len(sets) <= (5 if rz == 2 else 3)


def test_vae_subpixel_upsampler_in_the_engine(vae_setup):
    """Engine with the sub-pixel upsampler == engine with the reference's two steps up to the bf16 rounding of the merged
    weights (fp32 activations on CPU): untiled, temporally sliced (the causal memory is the low-resolution input's tail) and
    tiled."""
    cfg, sd, two_step = vae_setup
    merged = sub("vae").VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=torch.float32), merge_causal_head=False)
    assert all(up is None or up.merged is not None for _, up in merged.dec_up)
    assert all(up is None or up.merged is None for _, up in two_step.dec_up)
    assert sorted(len(up.merged) for _, up in merged.dec_up if up is not None) == [1, 5, 5]
    z = torch.randn(4, 6, 5, cfg.latent_channels, generator=torch.Generator().manual_seed(3)) * 0.5
    a, b = two_step.decode(z), merged.decode(z)
    assert a.shape == b.shape and 1e-5 < rel_err(b, a) < 6e-3
    for per_slice in (1, 2, 3):                                  # (fp32 torch convs: not bit-stable across shapes)
        assert rel_err(merged.decode(z, latents_per_slice=per_slice), b) < 1e-5
    assert rel_err(merged.decode(z[:1]), two_step.decode(z[:1])) < 6e-3          # a single latent frame: the head pattern only
    tile = dict(tiled=True, tile_size=(32, 32), tile_overlap=(8, 8))
    assert rel_err(merged.decode(z, **tile), two_step.decode(z, **tile)) < 4e-3


def test_vae_causal_head_merge_in_the_engine(vae_setup):
    """Output frame 0 of a clip computed as (W0+W1+W2) * x[0] with the sum held as two bf16 terms (the three temporal taps
    all fall on the replicated first frame: extend_head, causal_inflation_lib.py:422-437): the same function to 2^-17 per
    weight -- for encode and decode, any temporal slicing, single images, tiles, and together with the sub-pixel upsamplers."""
    cfg, sd, plain = vae_setup
    vae = sub("vae")
    ops = TorchOps("cpu", act_dtype=torch.float32)
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(3, 6, 5, cfg.latent_channels, generator=gen) * 0.5
    x = torch.randn(3, 9, 16, 24, generator=gen).clamp(-1, 1)
    m = vae.VideoVAEEngine(cfg, sd, ops, merge_upsamplers=False, merge_causal_head=True)
    c1 = m.dec_up[0][0][0].conv1
    assert c1.head is not None and c1.head.k == (2, 3, 3) and c1.head.head is None and plain.dec_up[0][0][0].conv1.head is None
    assert m.enc_conv_in.head is None and m.dec_conv_in.head is None and m.enc_down[0][1].head is None   # thin / strided: untouched
    w = sd["decoder.up_blocks.0.resnets.0.conv1.weight"].bfloat16().double()
    hl = c1.head.w[:w.shape[0]].double().reshape(w.shape[0], 2, 3, 3, w.shape[1]).sum(1).permute(0, 3, 1, 2)
    assert float((hl - w.sum(2)).abs().max()) <= 2.0 ** -16 * float(w.sum(2).abs().max())
    want = plain.decode(z)
    got = m.decode(z)
    assert rel_err(got, want) < 2e-5
    for per_slice in (1, 2):
        assert rel_err(m.decode(z, latents_per_slice=per_slice), got) < 1e-5
    assert rel_err(m.decode(z[:1]), plain.decode(z[:1])) < 2e-5                       # one latent frame: the head launch only
    assert rel_err(m.encode(x), plain.encode(x)) < 2e-5
    assert rel_err(m.encode(x, frames_per_slice=4), plain.encode(x)) < 2e-5
    assert rel_err(m.encode(x[:, :1]), plain.encode(x[:, :1])) < 2e-5
    tile = dict(tiled=True, tile_size=(32, 32), tile_overlap=(8, 8))
    assert rel_err(m.decode(z, **tile), plain.decode(z, **tile)) < 2e-5
    both = vae.VideoVAEEngine(cfg, sd, ops)                                           # the shipped default
    only_up = vae.VideoVAEEngine(cfg, sd, ops, merge_causal_head=False)
    assert rel_err(both.decode(z), only_up.decode(z)) < 2e-5 and rel_err(both.decode(z, latents_per_slice=1), both.decode(z)) < 1e-5


def test_vae_decode_skips_the_frames_the_caller_trims(vae_setup):
    """decode(z, keep_frames=n) == decode(z)[:, :n] for every n (the decoder is causal in time): untiled, one latent per slice,
    tiled; the latent frames that only feed trimmed output never enter the decoder, and at full frame rate the trimmed frames
    are not computed (fewer multiply-adds, counted through the ops)."""
    cfg, sd, eng = vae_setup
    z = torch.randn(4, 6, 5, cfg.latent_channels, generator=torch.Generator().manual_seed(9)) * 0.5
    variants = ({}, dict(latents_per_slice=1), dict(tiled=True, tile_size=(32, 32), tile_overlap=(8, 8)))
    full = [eng.decode(z, **kw) for kw in variants]
    assert full[0].shape[1] == 13
    for k in (1, 6, 9, 12, 20):                      # one frame; mid-clip cuts (whole latent frames dropped); the 4n+1 trim; no-op
        for kw, f in zip(variants, full):
            y = eng.decode(z, keep_frames=k, **kw)
            y = y.unsqueeze(1) if y.dim() == 3 else y
            assert y.shape[1] == min(k, 13) and rel_err(y, f[:, :k]) < 1e-5, (k, kw)
    with pytest.raises(ValueError):
        eng.decode(z, keep_frames=0)

    class Counting(TorchOps):
        macs, depth = 0.0, 0

        def gemm(self, A, W, out, *, N, K, M=None, conv=None, **kw):
            if self.depth == 0:
                self.macs += 2.0 * (conv.To * conv.Ho * conv.Wo if conv is not None else (M if M is not None else A.shape[0])) * N * K
            self.depth += 1
            try:
                return super().gemm(A, W, out, N=N, K=K, M=M, conv=conv, **kw)
            finally:
                self.depth -= 1

    ops = Counting("cpu", act_dtype=torch.float32)
    e2 = sub("vae").VideoVAEEngine(cfg, sd, ops)
    e2.decode(z)
    all13 = ops.macs
    e2.decode(z, keep_frames=12)
    cut12 = ops.macs - all13
    e2.decode(z, keep_frames=9)                       # the fourth latent frame is not needed at all
    cut9 = ops.macs - all13 - cut12
    assert cut9 < cut12 < all13 and cut12 > 0.9 * all13 and cut9 < 0.8 * all13


def test_flop_model_counts_what_the_engine_launches():
    """flops.py (bench.py's executed_tflop_per_step and the FLOPs behind roofline.achieved) against the multiply-adds of the
    GEMM / implicit-GEMM launches the engine really issues, for the reference's form and for both algebraic reductions
    (sub-pixel upsamplers, two-term causal head).  The only slack: the thin inputs are padded (encoder conv_in 3 -> 4
    channels, decoder conv_in K = 432 -> 448); what each reduction saves is compared exactly."""
    config, weights, vae, flops = sub("config"), sub("weights"), sub("vae"), sub("flops")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg)

    class Counting(TorchOps):
        macs, depth = 0.0, 0

        def gemm(self, A, W, out, *, N, K, M=None, conv=None, **kw):
            m = conv.To * conv.Ho * conv.Wo if conv is not None else (M if M is not None else A.shape[0])
            if self.depth == 0:                                   # (the torch restatement of a conv launch calls gemm again)
                self.macs += 2.0 * m * N * K
            self.depth += 1
            try:
                return super().gemm(A, W, out, N=N, K=K, M=M, conv=conv, **kw)
            finally:
                self.depth -= 1

    T, H, W = 9, 64, 64                                       # (64 latent voxels per frame: the mid attention runs as GEMMs)
    x = torch.rand(3, T, H, W) * 2 - 1
    z = torch.randn(3, H // 8, W // 8, cfg.latent_channels) * 0.5
    got, model = {}, {}
    for mu, mh in ((False, False), (True, False), (True, True)):
        ops = Counting("cpu", act_dtype=torch.float32)
        eng = vae.VideoVAEEngine(cfg, sd, ops, merge_upsamplers=mu, merge_causal_head=mh)
        eng.encode(x, frames_per_slice=4)
        enc = ops.macs
        eng.decode(z, latents_per_slice=1)
        got[mu, mh] = (enc, ops.macs - enc)
        f = flops.vae_flops_tiled(cfg, T, H, W, False, merged_upsamplers=mu, causal_head=mh)
        model[mu, mh] = (f["encode"], f["decode"])
        assert abs(got[mu, mh][0] / f["encode"] - 1) < 2e-3 and abs(got[mu, mh][1] / f["decode"] - 1) < 1e-5, (mu, mh, got[mu, mh], f)
    for a, b in (((False, False), (True, False)), ((True, False), (True, True))):      # what each reduction saves: exact
        for ph in (0, 1):
            assert abs((got[a][ph] - got[b][ph]) - (model[a][ph] - model[b][ph])) <= 1e-9 * model[a][ph]
    assert got[True, True][0] < got[True, False][0] and got[True, True][1] < got[True, False][1] < got[False, False][1]
    # decode(keep_frames=): the frames the caller trims are not computed, and the model counts the same launches
    for keep in (8, 5, 2):
        ops = Counting("cpu", act_dtype=torch.float32)
        vae.VideoVAEEngine(cfg, sd, ops).decode(z, latents_per_slice=1, keep_frames=keep)
        f = flops.vae_decode_flops(cfg, z.shape[0], H // 8, W // 8, True, True, keep_frames=keep)["total"]
        assert abs(ops.macs / f - 1) < 1e-4 and f < model[True, True][1], (keep, ops.macs, f)     # (slack: decoder conv_in's K pad)


def test_dit_engine_host_logic_matches_reference_golden():
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    g = torch.load(os.path.join(GOLDEN, "dit_tiny.pt"), weights_only=True)
    txt = torch.load(os.path.join(GOLDEN, "text_pos_emb.pt"), weights_only=True)
    eng = dit.NaDiTEngine(config.DIT_TINY, weights.synth_dit_state_dict(config.DIT_TINY), TorchOps("cpu", torch.float32))
    out = eng.forward(g["vid"].float(), txt.float(), 1000.0)
    assert rel_err(out, g["out"]) < 1e-5
    # reference-shaped call signature (NaDiT.forward)
    T, H, W, C = g["vid"].shape
    o2 = eng(g["vid"].float().reshape(-1, C), txt.float(), torch.tensor([[T, H, W]]), torch.tensor([[58]]),
             torch.tensor([1000.0])).vid_sample
    assert torch.equal(o2.reshape(T, H, W, -1), out)


def test_dit_engine_7b_family_matches_reference_golden_and_oracle():
    config, weights, dit, windows = sub("config"), sub("weights"), sub("dit"), sub("windows")
    cfg = config.DIT_7B_TINY
    g = torch.load(os.path.join(GOLDEN, "dit7b_tiny.pt"), weights_only=True)
    txt = torch.load(os.path.join(GOLDEN, "text_pos_emb.pt"), weights_only=True)
    eng = dit.NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg), TorchOps("cpu", torch.float32))
    assert rel_err(eng.forward(g["vid"].float(), txt.float(), 1000.0), g["out"]) < 1e-5
    sd = weights.synth_dit_state_dict(cfg, seed=11)                                 # ragged grid: shifted sliver windows
    eng = dit.NaDiTEngine(cfg, sd, TorchOps("cpu", torch.float32))
    torch.manual_seed(0)
    vid, t2 = torch.randn(5, 36, 60, 33), torch.randn(58, 5120)
    want = dit_oracle.dit_forward(sd, cfg, vid, t2, 1000.0, windows_mod=windows)
    assert rel_err(eng.forward(vid, t2, 1000.0), want) < 1e-5


def test_dit_engine_7b_production_width_host_logic_matches_reference_golden():
    """The 7B family's host logic at PRODUCTION width (3072 / 24 heads: 60 rotated dims, GELU MLP of 12288, rope tables per
    window extent) over the fp32 torch double of the C ABI == the imported reference's dit_7b code (tests/golden/dit7b_w2l_crop.pt)."""
    from oracle import make_golden as mg
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    g = torch.load(os.path.join(GOLDEN, "dit7b_w2l_crop.pt"), weights_only=True)
    txt = torch.load(os.path.join(GOLDEN, "text_pos_emb.pt"), weights_only=True)
    cfg = mg.dit7b_r3_config(config)
    eng = dit.NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg, seed=g["seed_weights"]), TorchOps("cpu", act_dtype=torch.float32))
    out = eng.forward(mg.dit_inputs(*g["latent"], seed=g["seed_input"]).float(), txt.float(), 1000.0)
    assert rel_err(out, g["out"]) < 2e-5


def test_dit_engine_ragged_grid_vs_oracle():
    config, weights, dit, windows = sub("config"), sub("weights"), sub("dit"), sub("windows")
    cfg = config.DIT_TINY
    sd = weights.synth_dit_state_dict(cfg, seed=11)
    eng = dit.NaDiTEngine(cfg, sd, TorchOps("cpu", torch.float32))
    torch.manual_seed(0)
    vid, txt = torch.randn(5, 36, 60, 33), torch.randn(58, 5120)
    want = dit_oracle.dit_forward(sd, cfg, vid, txt, 1000.0, windows_mod=windows)
    assert rel_err(eng.forward(vid, txt, 1000.0), want) < 1e-5


def test_vae_engine_encode_decode_slicing_tiling(vae_setup):
    cfg, sd, eng = vae_setup
    g = torch.load(os.path.join(GOLDEN, "vae_small.pt"), weights_only=True)
    tile = dict(tiled=True, tile_size=tuple(g["tile_size"]), tile_overlap=tuple(g["tile_overlap"]))
    x = g["x"][0].float()
    sc = cfg.scaling_factor
    assert rel_err(eng.encode(x), g["enc"][0].permute(1, 2, 3, 0) * sc) < 2e-5
    assert rel_err(eng.encode(x, **tile), g["enc_tiled"][0].permute(1, 2, 3, 0) * sc) < 2e-5
    z = g["z_in"][0].permute(1, 2, 3, 0).float() * sc
    assert rel_err(eng.decode(z, latents_per_slice=1), g["dec"][0]) < 2e-5
    assert rel_err(eng.decode(z, latents_per_slice=1, **tile), g["dec_tiled"][0]) < 2e-5


def test_vae_engine_17_frames_tiled_golden(vae_setup):
    """Round-2 fixture: 17 frames, 2x4 tiles + skipped tiles, the engine slicing at 4 frames / 1 latent like the reference
    and at its own budget-derived slice length."""
    from oracle import make_golden as mg
    cfg, sd, eng = vae_setup
    g = torch.load(os.path.join(GOLDEN, "vae_tiled17.pt"), weights_only=True)
    tile = dict(tiled=True, tile_size=tuple(g["tile_size"]), tile_overlap=tuple(g["tile_overlap"]))
    x = mg.blocky_frames(*g["frames"], seed=g["seed_x"], cell=g["cell"])[0].float()
    z = mg.latent_input(*g["latent"], seed=g["seed_z"])[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor
    want_e = g["enc_tiled"][0].permute(1, 2, 3, 0) * cfg.scaling_factor
    assert rel_err(eng.encode(x, frames_per_slice=4, **tile), want_e) < 2e-5
    assert rel_err(eng.encode(x, **tile), want_e) < 2e-5
    assert rel_err(eng.decode(z, latents_per_slice=1, **tile), g["dec_tiled"][0]) < 2e-5


def test_vae_engine_multi_slice_equals_oracle(vae_setup):
    cfg, sd, eng = vae_setup
    torch.manual_seed(0)
    x9 = torch.rand(3, 9, 32, 48) * 2 - 1
    want = vae_oracle.runner_vae_encode(x9, sd, cfg)
    assert rel_err(eng.encode(x9, frames_per_slice=4), want) < 2e-5       # 3 temporal slices, halos carried
    assert rel_err(eng.encode(x9), want) < 2e-5
    z = torch.randn(3, 4, 6, 16)
    want = vae_oracle.runner_vae_decode(z, sd, cfg)
    assert rel_err(eng.decode(z, latents_per_slice=1), want) < 2e-5
    img = torch.rand(3, 32, 32) * 2 - 1                                     # single image: [3, H, W]
    lat = eng.encode(img)
    assert lat.shape == (1, 4, 4, 16)
    assert eng.decode(lat).shape == (3, 32, 32)


def test_runner_api_surface():
    runner, config = sub("runner"), sub("config")
    r = runner.VideoDiffusionInfer(runner.default_config(config.DIT_TINY))
    lat = torch.randn(3, 4, 5, 16)
    cond = r.get_condition(lat, latent_blur=lat * 2, task="sr")
    assert cond.shape == (3, 4, 5, 17) and torch.equal(cond[..., :-1], lat * 2) and (cond[..., -1] == 1).all()
    r.configure_diffusion()
    t = r.timestep_transform(torch.tensor([1000.0]), torch.tensor([[9, 270, 480]]))
    assert abs(float(t) - 1000.0) < 1e-3                                   # t = T is a fixed point of the shift
    x = r.schedule.forward(torch.zeros(2, 2), torch.ones(2, 2), torch.tensor([250.0]))
    assert torch.allclose(x, torch.full((2, 2), 0.25))
    r.config.diffusion.timesteps.sampling.steps = 50                        # (round 6: any number of steps, trailing.py:39-48)
    r.configure_diffusion()
    ts = r.sampling_timesteps
    assert ts.shape == (50,) and float(ts[0]) == 1000.0 and abs(float(ts[1]) - 980.0) < 1e-3 and abs(float(ts[-1]) - 20.0) < 1e-3
    r.config.diffusion.timesteps.sampling.shift = 3.0                       # SD3 shift: t -> s t / (1 + (s - 1) t)
    r.configure_diffusion()
    assert abs(float(r.sampling_timesteps[25]) - 1000.0 * 3 * 0.5 / (1 + 2 * 0.5)) < 1e-3
    r.config.diffusion.sampler.prediction_type = "v_cos"                    # not a combination the reference's configs use
    with pytest.raises(NotImplementedError):
        r.configure_diffusion()
    r.config.diffusion.sampler.prediction_type = "v_lerp"
    r.config.diffusion.timesteps.sampling.steps = 1
    r.config.diffusion.timesteps.sampling.shift = 1.0
    r.configure_diffusion()
    assert r.inference([], [], [], []) == []


def test_conv_tile_order_is_a_bijection():
    """The conv kernel's tile id -> (frame, tile row, tile column, cout tile) decode (svr_conv_halo2.hip, frame index INSIDE a
    band of tile rows), restated: every tile is visited exactly once for ragged band counts, single frames, bands wider than
    the image, and the frame-outermost order (band 0); within a band the frames of one tile row are adjacent."""
    def decode(tl, tiles_n, tiles_x, tiles_y, To, band):
        tn, rr = tl % tiles_n, tl // tiles_n
        tx, rr = rr % tiles_x, rr // tiles_x
        if 0 < band < tiles_y:
            per_band = band * To
            b = rr // per_band
            rows_b = min(band, tiles_y - b * band)
            r2 = rr - b * per_band
            to = r2 // rows_b
            ty = b * band + (r2 - to * rows_b)
        else:
            ty, to = rr % tiles_y, rr // tiles_y
        return to, ty, tx, tn

    for tiles_n, tiles_x, tiles_y, To, band in ((1, 32, 64, 5, 1), (2, 3, 7, 4, 3), (4, 1, 10, 3, 4), (1, 5, 3, 9, 2),
                                                (1, 2, 5, 1, 4), (3, 2, 4, 2, 8), (1, 4, 6, 3, 0)):
        total = tiles_n * tiles_x * tiles_y * To
        seen = [decode(t, tiles_n, tiles_x, tiles_y, To, band) for t in range(total)]
        assert len(set(seen)) == total
        assert set(seen) == {(a, b, c, d) for a in range(To) for b in range(tiles_y) for c in range(tiles_x) for d in range(tiles_n)}
    # one-row bands: the To frames of a tile row occupy consecutive runs of tiles_x * tiles_n ids
    run = 32 * 1
    first = [decode(t * run, 1, 32, 64, 5, 1) for t in range(10)]
    assert [f[:2] for f in first] == [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (1, 1), (2, 1), (3, 1), (4, 1)]
