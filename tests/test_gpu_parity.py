"""-m gpu: end-to-end parity of the HIP engines against (a) the committed golden outputs of the REFERENCE implementation
(tests/golden, produced by oracle/make_golden.py from the imported reference in fp32) and (b) the in-repo CPU oracle run live
on the same seeded inputs.  Every test goes ctypes -> C ABI -> HIP kernels (``HipOps``); the oracle is only the checker.

Tolerances (the stated contract, SURVEY.md 8(c)(4) / DESIGN.md section 4; storage = bf16 MFMA operands + fp32 residual trunk /
stream).  A single bf16 rounding of a perfect result is 1.7e-3 L2 rel-err, so "1e-3" is enforced per kernel with an fp32
store (tests/test_gpu_kernels.py); end to end the asserts sit AT THE BAR of the north star:
  * decoded frames: PSNR >= 50.0 dB against the reference's fp32 output at the NOMINAL peak (2.0 for [-1, 1] frames, 1.0 for
    [0, 1] frames: ``psnr_nominal`` / ``_psnr_unit``) -- vae_tiled17, vae_tile1024, pipeline_small, pipeline_prod;
  * DiT predictions: rel-err <= 1e-2 at full depth (measured 4.6e-3 at 32 layers), PSNR against the output's own range;
  * wherever the fixture holds the REFERENCE's own bf16 run on the same inputs (refbf16.pt, pipeline_prod.pt): engine error
    <= reference-bf16 error -- never worse than what the reference does in its production dtype.
Measured values are printed by each test; the committed numbers live in DESIGN.md section 4.
"""
import math
import os

import pytest
import torch

from conftest import sub, rel_err, GOLDEN, ROOT as ROOT_DIR

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def psnr(a, b):
    rng = float(b.max() - b.min())
    mse = float((a.double() - b.double()).pow(2).mean())
    return 10 * math.log10(rng * rng / max(mse, 1e-30))


def psnr_nominal(a, b, peak=2.0):
    """PSNR of decoded frames against the nominal range of clamped [-1, 1] output (peak 2.0), SURVEY.md 7(ii).  With
    random-initialised weights the decoder's output spans about [-3.5, 3] (std 0.58), so this is the stricter measure."""
    mse = float((a.double() - b.double()).pow(2).mean())
    return 10 * math.log10(peak * peak / max(mse, 1e-30))


@pytest.fixture(scope="module")
def hip():
    return sub("ops").HipOps("cuda:0")


def _golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


def test_dit_tiny_vs_reference_golden(hip):
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    g, txt = _golden("dit_tiny.pt"), _golden("text_pos_emb.pt")
    eng = dit.NaDiTEngine(config.DIT_TINY, weights.synth_dit_state_dict(config.DIT_TINY, seed=g["seed_weights"]), hip)
    out = eng.forward(g["vid"].cuda(), txt.cuda(), 1000.0).float().cpu()
    assert rel_err(out, g["out"]) < 8e-3 and psnr(out, g["out"]) > 60      # measured 2.8e-3 / 70 dB


def test_dit_7b_family_vs_reference_golden(hip):
    """SeedVR2-7B graph (dit_7b: separate weights in every block, GELU MLP fused into the GEMM epilogue, pixel RoPE
    through the table-driven q/k-norm + RoPE kernel, no output norm) at reduced width."""
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    g, txt = _golden("dit7b_tiny.pt"), _golden("text_pos_emb.pt")
    cfg = config.DIT_7B_TINY
    eng = dit.NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg, seed=g["seed_weights"]), hip)
    out = eng.forward(g["vid"].cuda(), txt.cuda(), 1000.0).float().cpu()
    e, p = rel_err(out, g["out"]), psnr(out, g["out"])
    print(f"DiT-7B family (tiny): rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert e < 8e-3 and p > 60                          # measured 2.7e-3 / 70 dB


_ENGINES = {}


def _dit3b_engine(hip, seed):
    """The full SeedVR2-3B engine (32 layers, 3.4e9 synthetic parameters drawn on the CPU generator like the fixtures): built
    once per test session, shared by the tests that need it."""
    if ("3b", seed) not in _ENGINES:
        config, weights, dit = sub("config"), sub("weights"), sub("dit")
        _ENGINES["3b", seed] = dit.NaDiTEngine(config.DIT_3B, weights.synth_dit_state_dict(config.DIT_3B, seed=seed), hip)
    return _ENGINES["3b", seed]


def test_dit_3b_cfg1_vs_reference_golden(hip):
    """BASELINE config 1 shape (latent 1x32x32), full 32-layer SeedVR2-3B, synthetic weights."""
    g, txt = _golden("dit3b_cfg1.pt"), _golden("text_pos_emb.pt")
    eng = _dit3b_engine(hip, g["seed_weights"])
    out = eng.forward(g["vid"].cuda(), txt.cuda(), 1000.0).float().cpu()
    e, p = rel_err(out, g["out"]), psnr(out, g["out"])
    print(f"DiT-3B cfg-1: rel-err {e:.3e} (reference bf16 path: 1.66e-2; round 2 with a bf16 residual stream: 1.45e-2), PSNR {p:.1f} dB")
    assert e < 1.0e-2 and p > 55                       # measured 4.6e-3 / 64.4 dB with the fp32 residual stream


def test_dit_3b_full_depth_multiwindow_vs_reference_golden(hip):
    """FULL depth at production width on a multi-window grid: all 32 layers (10 MM + 22 shared, last block video-only) of
    SeedVR2-3B on the cropped BASELINE config-3 grid 9x30x54 tokens -- 12 regular / 36 shifted ragged windows in alternating
    layers (window.py:51-83, mmattn.py:161-271); golden from the imported reference in fp32 (11 CPU-minutes,
    oracle/make_golden.py --only r3-dit32)."""
    from oracle import make_golden as mg
    g, txt = _golden("dit3b_32l_crop.pt"), _golden("text_pos_emb.pt")
    eng = _dit3b_engine(hip, g["seed_weights"])
    vid = mg.dit_inputs(*g["latent"], seed=g["seed_input"]).cuda()
    out = eng.forward(vid, txt.cuda(), 1000.0).float().cpu()
    e, p = rel_err(out, g["out"]), psnr(out, g["out"])
    print(f"DiT-3B, 32 layers, 48 windows per layer pair (9x30x54 tokens): rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert e < 1.0e-2 and p > 55


def test_pipeline_cfg1_in_full_vs_reference_golden(hip):
    """BASELINE config 1 IN FULL, end to end against the reference (round 6): one 128 x 128 image -> 256 x 256 through the whole
    chain -- input transform, UNTILED VAE encode (latent 1 x 32 x 32), the 32-layer SeedVR2-3B NaDiT over 256 video + 58 text tokens,
    one Euler step, VAE decode, LAB colour fix -- golden = the reference's own model classes and glue in fp32, plus the reference's
    own bf16 run (oracle/make_golden.py --only r6-cfg1).  infer.py:117,203,315; generation_phases.py:171,542,807,1060."""
    from oracle import make_golden as mg
    config, weights, vae, runner, pipeline = (sub(n) for n in ("config", "weights", "vae", "runner", "pipeline"))
    g = _golden("pipeline_cfg1.pt")
    dcfg, vcfg = getattr(config, g["dit"]), config.VAEConfig(block_out_channels=tuple(g["vae_channels"]))
    assert dcfg is config.DIT_3B and vcfg.block_out_channels == config.VAE_V3.block_out_channels
    r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg))
    r.dit = _dit3b_engine(hip, g["seed_dit"])
    r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=g["seed_vae"]), hip)
    images = torch.rand(g["frames"], g["hw"][0], g["hw"][1], 3, generator=torch.Generator().manual_seed(g["seed_images"]))
    out = pipeline.upscale(images.cuda(), r, weights.synth_text_embedding().cuda(), resolution=g["resolution"],
                           batch_size=g["batch_size"], uniform_batch_size=g["uniform_batch_size"],
                           temporal_overlap=g["temporal_overlap"], color_correction="lab",
                           noise_provider=mg.pipeline_noise).float().cpu()
    assert out.shape == g["out"].shape == (1, 256, 256, 3)
    p, e = _psnr_unit(out, g["out"]), rel_err(out, g["out"])
    p_ref = _psnr_unit(g["out_refbf16"].float(), g["out"])
    print(f"BASELINE config 1 in full (256x256 image, 3B x 32 layers + full VAE) vs the reference chain: PSNR {p:.1f} dB at the nominal "
          f"peak, rel-err {e:.3e}; the reference chain in bf16 on the same input: {p_ref:.1f} dB")
    assert p >= 50.0 and p >= p_ref


def test_pipeline_production_width_and_depth_vs_reference_golden(hip):
    """The WHOLE chain at production width and depth (round 4): the reference's own four phases
    (generation_phases.py:171,542,807,1060, restated by oracle/pipeline_oracle.py and pinned to the phase text by
    tests/test_reference_phases_dropin.py) over its 32-layer SeedVR2-3B NaDiT (dit_3b/nadit.py:190) and its full-width VAE
    (attn_video_vae.py:1660; 128-256-512-512) -- 9 frames 48x80 -> 96x160, batches of 5 with uniform padding and a 2-frame
    overlap blend, VAE tiled 64 / 16 px in encode and decode (per-tile GroupNorm statistics and attention, cosine seams), LAB
    colour fix; golden in fp32 plus the reference's own bf16 run of the same chain (oracle/make_golden.py --only r4-prod)."""
    from oracle import make_golden as mg
    config, weights, vae, runner, pipeline = (sub(n) for n in ("config", "weights", "vae", "runner", "pipeline"))
    g = _golden("pipeline_prod.pt")
    dcfg, vcfg = getattr(config, g["dit"]), config.VAEConfig(block_out_channels=tuple(g["vae_channels"]))
    assert dcfg is config.DIT_3B and vcfg.block_out_channels == config.VAE_V3.block_out_channels
    r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg), encode_tiled=True, decode_tiled=True,
                                   encode_tile_size=tuple(g["vae_tile"]), encode_tile_overlap=tuple(g["vae_tile_overlap"]),
                                   decode_tile_size=tuple(g["vae_tile"]), decode_tile_overlap=tuple(g["vae_tile_overlap"]))
    r.dit = _dit3b_engine(hip, g["seed_dit"])
    r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=g["seed_vae"]), hip)
    images = torch.rand(g["frames"], g["hw"][0], g["hw"][1], 3, generator=torch.Generator().manual_seed(g["seed_images"]))
    out = pipeline.upscale(images.cuda(), r, weights.synth_text_embedding().cuda(), resolution=g["resolution"],
                           batch_size=g["batch_size"], uniform_batch_size=g["uniform_batch_size"],
                           temporal_overlap=g["temporal_overlap"], color_correction="lab",
                           noise_provider=mg.pipeline_noise).float().cpu()
    assert out.shape == g["out"].shape
    p, e = _psnr_unit(out, g["out"]), rel_err(out, g["out"])
    p_ref = _psnr_unit(g["out_refbf16"].float(), g["out"])
    print(f"pipeline at production width/depth (3B x 32 layers + full VAE, tiled) vs the reference chain: PSNR {p:.1f} dB at the "
          f"nominal peak, rel-err {e:.3e}; the reference chain in bf16 on the same inputs: {p_ref:.1f} dB")
    assert p >= 50.0 and p >= p_ref
    _ENGINES.clear()                                   # (7 GB of weights: not needed by the tests that follow)


def test_dit_7b_full_depth_vs_reference_golden(hip):
    """SeedVR2-7B at FULL depth and production width: 36 layers x 3072 (8.2e9 synthetic parameters; dit_7b/nadit.py,
    configs_7b/main.yaml:11-33) at BASELINE config 1's shape (latent 1x32x32); golden from the imported reference's dit_7b code
    in fp32 (33 GB of weights on the build box, oracle/make_golden.py --only r4-dit7b36)."""
    from oracle import make_golden as mg
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    g, txt = _golden("dit7b_36l_cfg1.pt"), _golden("text_pos_emb.pt")
    cfg = config.DIT_7B
    eng = dit.NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg, seed=g["seed_weights"]), hip)
    vid = mg.dit_inputs(*g["latent"], seed=g["seed_input"]).cuda()
    out = eng.forward(vid, txt.cuda(), 1000.0).float().cpu()
    e, p = rel_err(out, g["out"]), psnr(out, g["out"])
    print(f"DiT-7B, 36 layers at production width (cfg-1 shape): rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert e < 1.0e-2 and p > 55
    del eng
    torch.cuda.empty_cache()


def test_dit_7b_production_width_vs_reference_golden(hip):
    """SeedVR2-7B at PRODUCTION width (3072, 24 heads x 128, 60 rotated dims per head from the "pixel" RoPE, biased GELU MLP
    of 12288, separate vid / txt weights, no output norm: dit_7b/nablocks/mmsr_block.py:33-160), 2 layers = one regular and
    one shifted window layer, 5x30x54 tokens; golden from the imported reference's dit_7b code in fp32."""
    from oracle import make_golden as mg
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    g, txt = _golden("dit7b_w2l_crop.pt"), _golden("text_pos_emb.pt")
    cfg = mg.dit7b_r3_config(config)
    eng = dit.NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg, seed=g["seed_weights"]), hip)
    vid = mg.dit_inputs(*g["latent"], seed=g["seed_input"]).cuda()
    out = eng.forward(vid, txt.cuda(), 1000.0).float().cpu()
    e, p = rel_err(out, g["out"]), psnr(out, g["out"])
    print(f"DiT-7B production width, 2 layers: rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert e < 6e-3 and p > 60


def test_dit_3b_width_multiwindow_vs_reference_golden(hip):
    """Production WIDTH (2560, 20 heads x 128) on a cropped BASELINE config-3 token grid 9x30x54: the config-3 window
    (3x15x27 = 1215 video rows + 58 text rows), 12 regular + 36 shifted windows with ragged edges (77 .. 1215 rows),
    2 MM + 2 shared blocks, last block video-only; golden from the imported reference (fp32).  Both window-attention
    kernels.  Measured on MI355X: 2.9e-3 / 70 dB; asserted: <= 9e-3, and never worse than the reference's own bf16 run."""
    from oracle import make_golden as mg
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    g, txt = _golden("dit3b_w4l_crop.pt"), _golden("text_pos_emb.pt")
    cfg = mg.dit_r2_config(config)
    eng = dit.NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg, seed=g["seed_weights"]), hip)
    vid = mg.dit_inputs(*g["latent"], seed=g["seed_input"]).cuda()
    outs = {}
    for impl in (0, 1):
        hip.set_option("attn_impl", impl)
        try:
            out = eng.forward(vid, txt.cuda(), 1000.0).float().cpu()
        finally:
            hip.set_option("attn_impl", 0)
        e, p = rel_err(out, g["out"]), psnr(out, g["out"])
        print(f"DiT 3B-width 4 layers, 48 windows, attn_impl={impl}: rel-err {e:.3e}, PSNR {p:.1f} dB")
        assert e < 9e-3 and p > 60, (impl, e, p)      # measured 2.9e-3 / 70.3 dB with the fp32 residual stream
        outs[impl] = out
    assert rel_err(outs[0], outs[1]) < 4e-3
    e_ref = rel_err(_golden("refbf16.pt")["dit3b_w4l_crop"].float(), g["out"])
    print(f"reference bf16 path on the same inputs: rel-err {e_ref:.3e}")
    assert rel_err(outs[0], g["out"]) <= e_ref       # never worse than what the reference itself does in its production dtype


def test_dit_runner_euler_endpoint(hip):
    """runner.inference == x_t - dit(x_t || cond) (one-step Euler endpoint fused into un-patchify)."""
    config, weights, dit, runner = sub("config"), sub("weights"), sub("dit"), sub("runner")
    g, txt = _golden("dit_tiny.pt"), _golden("text_pos_emb.pt")
    eng = dit.NaDiTEngine(config.DIT_TINY, weights.synth_dit_state_dict(config.DIT_TINY, seed=g["seed_weights"]), hip)
    r = runner.VideoDiffusionInfer(runner.default_config(config.DIT_TINY))
    r.dit = eng
    r.configure_diffusion()
    vid = g["vid"].cuda()
    noise, cond = vid[..., :16].contiguous(), vid[..., 16:].contiguous()
    x0 = r.inference([noise], [cond], [txt.cuda()], [txt.cuda()])[0].float().cpu()
    want = g["vid"][..., :16].float() - g["out"]
    assert rel_err(x0, want) < 2.0e-2


@pytest.mark.parametrize("tiled", [False, True])
def test_vae_vs_reference_golden(hip, tiled):
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    g = _golden("vae_small.pt")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, seed=g["seed_weights"]), hip)
    kw = dict(tiled=True, tile_size=tuple(g["tile_size"]), tile_overlap=tuple(g["tile_overlap"])) if tiled else {}
    lat = eng.encode(g["x"][0].cuda(), **kw).float().cpu()
    want = g["enc_tiled" if tiled else "enc"][0].permute(1, 2, 3, 0) * cfg.scaling_factor
    e, p = rel_err(lat, want), psnr(lat, want)
    print(f"VAE encode tiled={tiled}: rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert e < 2.5e-2 and p > 50
    z = (g["z_in"][0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(BF16).cuda()
    y = eng.decode(z, **kw).float().cpu()
    want = g["dec_tiled" if tiled else "dec"][0]
    e, p = rel_err(y, want), psnr(y, want)
    print(f"VAE decode tiled={tiled}: rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert e < 2.5e-2 and p > 50


def test_vae_17_frames_multitile_vs_reference_golden(hip):
    """17 frames (5 temporal slices in the reference's split-4 scheme; here one slice AND forced 4-frame slices),
    2x4 spatial tiles with the skipped-tile rule exercised on both axes; golden from the imported reference."""
    from oracle import make_golden as mg
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    g = _golden("vae_tiled17.pt")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, seed=g["seed_weights"]), hip)
    kw = dict(tiled=True, tile_size=tuple(g["tile_size"]), tile_overlap=tuple(g["tile_overlap"]))
    x = mg.blocky_frames(*g["frames"], seed=g["seed_x"], cell=g["cell"])[0].cuda()
    want = g["enc_tiled"][0].permute(1, 2, 3, 0) * cfg.scaling_factor
    lat = eng.encode(x, **kw)
    assert torch.equal(lat, eng.encode(x, frames_per_slice=4, **kw))                 # slicing is bit-invisible
    e, p = rel_err(lat.float().cpu(), want), psnr(lat.float().cpu(), want)
    print(f"VAE encode 17 frames, 2x4 tiles: rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert e < 2.5e-2 and p > 50
    z = (mg.latent_input(*g["latent"], seed=g["seed_z"])[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(BF16).cuda()
    y = eng.decode(z, **kw)
    assert torch.equal(y, eng.decode(z, latents_per_slice=1, **kw))
    y = y.float().cpu()
    e, p, pn = rel_err(y, g["dec_tiled"][0]), psnr(y, g["dec_tiled"][0]), psnr_nominal(y, g["dec_tiled"][0])
    print(f"VAE decode 17 frames, 2x4 tiles: rel-err {e:.3e}, PSNR {p:.1f} dB (own range), {pn:.1f} dB (nominal peak 2.0)")
    assert e < 2.5e-2 and p > 50 and pn >= 50.0        # the north star's bar at the nominal peak (round 2, bf16 trunk: 49.3 dB)
    rb = _golden("refbf16.pt")                          # the reference's own bf16 run on the same inputs: the engine must not be worse
    e_ref = rel_err(rb["vae_tiled17_dec"][0].float(), g["dec_tiled"][0])
    e_ref_enc = rel_err(rb["vae_tiled17_enc"][0].float().permute(1, 2, 3, 0) * cfg.scaling_factor, want)
    print(f"reference bf16 path on the same inputs: encode rel-err {e_ref_enc:.3e}, decode rel-err {e_ref:.3e} "
          f"({psnr_nominal(rb['vae_tiled17_dec'][0].float(), g['dec_tiled'][0]):.1f} dB nominal)")
    assert e <= e_ref and rel_err(lat.float().cpu(), want) <= e_ref_enc


def test_vae_real_tile_size_strip_vs_reference_golden(hip):
    """The headline tile geometry: 5 frames 1024x1152 = a 2-tile strip at tile 1024 / overlap 128 (what BASELINE config 3
    runs 15 times per batch), golden from the imported reference (fp32, 9 min of CPU): encode compared whole, decode on 8
    96x96 crops incl. the blend seam (x 896..1024), corners and tile interiors."""
    from oracle import make_golden as mg
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    g = _golden("vae_tile1024.pt")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, seed=g["seed_weights"]), hip)
    kw = dict(tiled=True, tile_size=tuple(g["tile_size"]), tile_overlap=tuple(g["tile_overlap"]))
    x = mg.blocky_frames(*g["frames"], seed=g["seed_x"], cell=g["cell"])[0].cuda()
    lat = eng.encode(x, **kw).float().cpu()
    want = g["enc_tiled"][0].permute(1, 2, 3, 0) * cfg.scaling_factor
    e, p = rel_err(lat, want), psnr(lat, want)
    print(f"VAE encode 1024-px tiles (2-tile strip): rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert e < 2.5e-2 and p > 50
    del x
    z = (mg.latent_input(*g["latent"], seed=g["seed_z"])[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(BF16).cuda()
    y = eng.decode(z, **kw).float().cpu()                                            # [3, 5, 1024, 1152]
    assert y.shape == (3, 5, 1024, 1152)
    got = torch.stack([y[:, :, yy:yy + 96, xx:xx + 96] for (yy, xx) in g["crops"]])
    e, p, pn = rel_err(got, g["dec_crops"]), psnr(got, g["dec_crops"]), psnr_nominal(got, g["dec_crops"])
    print(f"VAE decode 1024-px tiles (2-tile strip, 8 crops): rel-err {e:.3e}, PSNR {p:.1f} dB (own range), "
          f"{pn:.1f} dB (nominal peak 2.0)")
    assert e < 2.5e-2 and p > 50 and pn >= 50.0        # (round 2, bf16 trunk: 50.3 dB)
    assert abs(float(y.mean()) - g["dec_mean"]) < 5e-3 and abs(float(y.std()) - g["dec_std"]) < 5e-3


@pytest.mark.parametrize("name", ["vae_untiled_2048", "vae_untiled_1024x5"])
def test_vae_untiled_cfg2_geometry_vs_reference_golden(hip, name):
    """BASELINE config 2's GEOMETRY against the reference itself (round 6; before, this size was checked per op against this repo's own
    torch restatement only): the UNTILED VAE on one 2048 x 2048 frame -- per-frame GroupNorm groups over 4.2e6 pixels
    (causal_inflation_lib.py:354-409), 65 536-token mid-block attention (attn_video_vae.py:615-665) -- and on a 5-frame 1024 x 1024
    clip (two temporal slices through the reference's causal memory, attn_video_vae.py:1254-1300; 16 384-token attention).  Goldens
    from the imported reference in fp32 (oracle/make_golden.py --only r6-vaebig; 28 / 25 CPU-minutes); encode compared whole, decode
    on eight 96 x 96 crops (corners, centre, interior) plus the mean / std of the whole output."""
    from oracle import make_golden as mg
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    g = _golden(name + ".pt")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, seed=g["seed_weights"]), hip)
    x = mg.blocky_frames(*g["frames"], seed=g["seed_x"], cell=g["cell"])[0].cuda()
    lat = eng.encode(x).float().cpu()
    lat = lat[None] if lat.dim() == 3 else lat
    want = g["enc"][0].permute(1, 2, 3, 0) * cfg.scaling_factor
    e, p = rel_err(lat, want), psnr(lat, want)
    print(f"{name}: untiled VAE encode {tuple(g['frames'])}: rel-err {e:.3e}, PSNR {p:.1f} dB")
    assert lat.shape == want.shape and e < 2.5e-2 and p > 50
    del x
    z = (mg.latent_input(*g["latent"], seed=g["seed_z"])[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(BF16).cuda()
    y = eng.decode(z).float().cpu()
    y = y.unsqueeze(1) if y.dim() == 3 else y                                        # [3, T, H, W]
    assert y.shape == (3,) + tuple(g["frames"])
    got = torch.stack([y[:, :, yy:yy + 96, xx:xx + 96] for (yy, xx) in g["crops"]])
    e, p, pn = rel_err(got, g["dec_crops"]), psnr(got, g["dec_crops"]), psnr_nominal(got, g["dec_crops"])
    print(f"{name}: untiled VAE decode (8 crops): rel-err {e:.3e}, PSNR {p:.1f} dB (own range), {pn:.1f} dB (nominal peak 2.0)")
    assert e < 2.5e-2 and p > 50 and pn >= 50.0
    assert abs(float(y.mean()) - g["dec_mean"]) < 5e-3 and abs(float(y.std()) - g["dec_std"]) < 5e-3


@pytest.mark.parametrize("level", ["tail", "overflow"])
def test_heavy_tail_checkpoint_statistics_on_the_hip_path(hip, level):
    """Round 6: the h16 regime and its overflow guards on weights with outlier channels (x300 in the NaDiT's residual stream),
    GroupNorm gains / modulation scales over three decades, O(1) biases, a 10x conv and e4m3-valued matrices, against the reference's
    fp32 run (yardstick: the reference's own bf16 run).  "tail": guards silent, engine error <= reference-bf16 error.  "overflow": the
    stream / trunk leaves h16's range -> the guard re-runs the call with fp32 stores ON THE HIP PATH, warns, and the result still
    matches.  Bodies shared with the CPU-double test (tests/heavy_tail_cases.py)."""
    import heavy_tail_cases as ht
    ht.dit_case(hip, level)
    ht.vae_case(hip, level)


def test_multistep_euler_with_classifier_free_guidance_on_the_hip_path(hip):
    """Round 6: runner.inference beyond the pipeline's forced steps = 1 / cfg = 1 -- four trailing Euler steps, classifier-free guidance
    (scale 2.5 on the first half of the steps, rescale 0.7), two clips of different sizes -- against the reference's own EulerSampler /
    CFG dispatcher over its NaDiT (tests/golden/sampler_multistep.pt; euler.py:36-102, diffusion/utils.py:41-86, infer.py:315-395).
    Body shared with the CPU-double test; six model calls per clip accumulate, the bound is 3x a single call's."""
    import test_sampler_multistep as sm
    errs = sm.run_case(hip) + sm.run_case(hip, plain=True)
    print("multi-step Euler + CFG on the HIP path vs the reference's sampler:", ["%.2e" % e for e in errs])
    assert max(errs) < 1.5e-2


def test_vae_temporal_slicing_invariance(hip):
    """Size-independent property: slice size must not change the result (causal halos carry state).
    Every kernel on the path reduces in a fixed order (no atomics), so the property is checked BIT-EXACT:
    a random-weight VAE amplifies any perturbation to the bf16 noise floor (1e-2) within ~20 layers, so a
    tolerance could not tell a halo bug from rounding noise, equality can."""
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg), hip)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(3, 13, 64, 64, generator=g) * 2 - 1).to(BF16).cuda()
    a, b = eng.encode(x), eng.encode(x, frames_per_slice=4)
    assert torch.equal(a, b), rel_err(b.float(), a.float())
    assert torch.equal(a, eng.encode(x, frames_per_slice=8))
    z = torch.randn(4, 8, 8, 16, generator=g).to(BF16).cuda()
    a, b = eng.decode(z), eng.decode(z, latents_per_slice=1)
    assert a.shape == (3, 13, 64, 64) and torch.equal(a, b), rel_err(b.float(), a.float())


def test_vae_tile_streams_are_bit_invisible(hip):
    """Spatial tiles issued round-robin on 2 or 3 HIP streams (VideoVAEEngine(tile_streams=), vae.py::_run_tiles) against every
    launch on one stream: same launches, same blend order -> the same bits, encode and decode, also when the run is repeated
    (a race between a tile's launches and another tile's allocations would show up as a flipped value somewhere)."""
    from oracle import make_golden as mg
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg)
    kw = dict(tiled=True, tile_size=(64, 64), tile_overlap=(16, 16))
    x = mg.blocky_frames(9, 160, 224, seed=51, cell=8)[0].cuda()                     # 3 x 4 tiles of 64 px (stride 48)
    z = (mg.latent_input(3, 20, 28, seed=52)[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(BF16).cuda()
    ref_eng = vae.VideoVAEEngine(cfg, sd, hip, tile_streams=1)
    lat1, dec1 = ref_eng.encode(x, **kw), ref_eng.decode(z, **kw)
    for n in (2, 3):
        eng = vae.VideoVAEEngine(cfg, sd, hip, tile_streams=n)
        for rep in range(2):
            assert torch.equal(eng.encode(x, **kw), lat1), (n, rep)
            assert torch.equal(eng.decode(z, **kw), dec1), (n, rep)
            assert torch.equal(eng.decode(z, keep_frames=6, **kw), dec1[:, :6]), (n, rep)
    torch.cuda.synchronize()


def test_vae_decode_keep_frames_bit_exact(hip):
    """VideoVAEEngine.decode(keep_frames=n) -- what pipeline.upscale asks for by default so that the 4n+1 / uniform-batch padding it
    trims is never decoded -- equals the first n frames of the full decode BIT FOR BIT on the HIP path: untiled, forced one-latent
    slices, spatially tiled; every cut position incl. "nothing to trim"."""
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, device="cuda"), hip)
    g = torch.Generator(device="cuda").manual_seed(11)
    z = (torch.randn(4, 12, 10, cfg.latent_channels, generator=g, device="cuda") * 0.5).to(BF16)
    for kw in ({}, dict(latents_per_slice=1), dict(tiled=True, tile_size=(64, 64), tile_overlap=(16, 16))):
        full = eng.decode(z, **kw)
        for k in (1, 2, 5, 9, 12, 13):
            y = eng.decode(z, keep_frames=k, **kw)
            y = y.unsqueeze(1) if y.dim() == 3 else y
            assert y.shape[1] == k and torch.equal(y, full[:, :k]), (kw, k)


def test_vae_oracle_live_ragged(hip):
    """Live CPU-oracle comparison on a ragged (non multiple-of-tile) size with 3 tiles per axis."""
    from oracle import vae_oracle
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg)
    eng = vae.VideoVAEEngine(cfg, sd, hip)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2, 9, 11, 16, generator=g).to(BF16)
    kw = dict(tiled=True, tile_size=(32, 40), tile_overlap=(8, 16))
    want = vae_oracle.runner_vae_decode(z.float(), sd, cfg, **kw)
    got = eng.decode(z.cuda(), **kw).float().cpu()
    assert got.shape == want.shape and rel_err(got, want) < 2.5e-2


def test_dit_window_attention_properties_full_size(hip):
    """BASELINE config-2 token grid (3 x 128 x 128 -> 49 152 tokens, 147 windows), one 3B-width block:
    (i) permutation property -- attention output must not depend on window enumeration order is implied by
    (ii) text rows see identical K/V in every window: shuffling the video rows inside a window leaves the
    text output of that window unchanged;  checked through the kernel at full width (20 heads)."""
    windows = sub("windows")
    plan = windows.plan_windows((3, 128, 128), (4, 3, 3), windows.SHIFTED)
    N, Lt, heads, D = 3 * 128 * 128, 58, 20, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = (torch.randn(N + Lt, 3 * heads * D, generator=g, device="cuda") * 0.5).to(BF16)
    w = plan.n_win // 2
    rows = torch.from_numpy(plan.tok[plan.cu[w]:plan.cu[w + 1]].copy()).cuda().to(torch.int32)
    txt = torch.arange(N, N + Lt, dtype=torch.int32, device="cuda")
    perm = rows[torch.randperm(rows.numel(), device="cuda")]
    outs = []
    for r in (rows, perm):
        seq = torch.cat([r, txt]).contiguous()
        L = seq.numel()
        dst = torch.arange(L, dtype=torch.int32, device="cuda")
        cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
        out = torch.zeros(L, heads * D, device="cuda", dtype=BF16)
        hip.attn_varlen(qkv, out, seq, dst, cu, L, heads, D, 1 / math.sqrt(D))
        outs.append(out[-Lt:].float())
    assert rel_err(outs[1], outs[0]) < 4e-3


def _psnr_unit(a, b):
    """PSNR of [0, 1] output frames against their nominal range (peak 1.0 = the 2.0 of [-1, 1] frames)."""
    mse = float((a.double() - b.double()).pow(2).mean())
    return 10 * math.log10(1.0 / max(mse, 1e-30))


def test_pipeline_vs_reference_golden(hip):
    """The whole chain (batching with uniform padding, 4n+1 padding, input transform in bf16, VAE encode, one-step DiT, VAE
    decode, trims, 2-frame overlap blend, LAB colour fix, [-1,1] -> [0,1]) on the GPU against the golden of the REFERENCE's
    components in fp32 (tests/golden/pipeline_small.pt, oracle/make_golden.py --only r2-pipe) -- not against this repo's
    own host logic.  Measured on MI355X (round 3, fp32 trunk / stream): 50.7 dB at the nominal peak (1.0 on [0, 1] frames),
    rel-err 5.2e-3; the reference's own chain in bf16 on the same inputs: 47.3 dB.  Asserted at the bar: >= 50.0 dB and
    >= the reference-bf16 run (tests/golden/refbf16.pt).  The same chain at production width and depth:
    test_pipeline_production_width_and_depth_vs_reference_golden."""
    from oracle import make_golden as mg
    config, weights, dit, vae, runner, pipeline = (sub(n) for n in ("config", "weights", "dit", "vae", "runner", "pipeline"))
    g = _golden("pipeline_small.pt")
    dcfg, vcfg = config.DIT_TINY, config.VAEConfig(block_out_channels=tuple(g["vae_channels"]))
    r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg))
    r.dit = dit.NaDiTEngine(dcfg, weights.synth_dit_state_dict(dcfg, seed=g["seed_dit"]), hip)
    r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=g["seed_vae"]), hip)
    images = torch.rand(g["frames"], g["hw"][0], g["hw"][1], 3, generator=torch.Generator().manual_seed(g["seed_images"]))
    out = pipeline.upscale(images.cuda(), r, weights.synth_text_embedding().cuda(), resolution=g["resolution"],
                           batch_size=g["batch_size"], uniform_batch_size=g["uniform_batch_size"],
                           temporal_overlap=g["temporal_overlap"], color_correction="lab",
                           noise_provider=mg.pipeline_noise).float().cpu()
    assert out.shape == g["out"].shape
    p, e = _psnr_unit(out, g["out"]), rel_err(out, g["out"])
    print(f"pipeline GPU vs reference-chain golden: PSNR {p:.1f} dB (nominal peak), rel-err {e:.3e}")
    rb = _golden("refbf16.pt")["pipeline_small"].float()
    p_ref = _psnr_unit(rb, g["out"])
    print(f"reference chain in bf16 on the same inputs: PSNR {p_ref:.1f} dB, rel-err {rel_err(rb, g['out']):.3e}")
    assert p >= 50.0 and e < 8e-3 and p >= p_ref       # the north star's bar end to end; never worse than the reference's bf16 path


def test_vae_full_tile_size_properties(hip):
    """BASELINE cfg3's working size (one 1024-px tile, 17 frames): properties that do not need an oracle run --
    temporal slicing is bit-invisible through the LDS-halo conv kernels, fused GroupNorm statistics and attention;
    the encoder is deterministic; latent scaling is the only difference between encode() and encode_clip()."""
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, device="cuda"), hip)
    g = torch.Generator(device="cuda").manual_seed(9)
    x = (torch.rand(3, 17, 1024, 1024, generator=g, device="cuda") * 2 - 1).to(BF16)
    a = eng.encode(x)
    assert a.shape == (5, 128, 128, 16) and bool(torch.isfinite(a.float()).all())
    assert torch.equal(a, eng.encode(x, frames_per_slice=8)) and torch.equal(a, eng.encode(x))
    z = (torch.randn(5, 64, 64, 16, generator=g, device="cuda")).to(BF16)        # 512-px tile through the decoder
    d = eng.decode(z)
    assert d.shape == (3, 17, 512, 512) and torch.equal(d, eng.decode(z, latents_per_slice=2))


def _nccl_worker(rank, world, port, q):
    import sys as _sys
    _sys.path.insert(0, ROOT_DIR)
    _sys.path.insert(0, os.path.join(ROOT_DIR, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from conftest import sub as _sub
    import torch.distributed as _dist
    d = _sub("dist")
    d.init_from_env(backend="nccl")
    torch.cuda.set_device(rank)
    config, weights, dit, vae, runner, pipeline = (_sub(n) for n in ("config", "weights", "dit", "vae", "runner", "pipeline"))
    ops = _sub("ops").HipOps(f"cuda:{rank}")
    dcfg, vcfg = config.DIT_TINY, config.VAEConfig(block_out_channels=(128, 128, 128, 128))
    r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg))
    r.dit = dit.NaDiTEngine(dcfg, weights.synth_dit_state_dict(dcfg, seed=21), ops)
    r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=22), ops)
    images = torch.rand(23, 24, 40, 3, generator=torch.Generator().manual_seed(4)).cuda(rank)
    kw = dict(resolution=48, batch_size=5, uniform_batch_size=True, temporal_overlap=2, color_correction="lab")
    out = d.upscale_sharded(images, r, weights.synth_text_embedding().cuda(rank), **kw)
    want = pipeline.upscale(images, r, weights.synth_text_embedding().cuda(rank), **kw) if rank == 0 else None
    q.put((rank, out.float().cpu().numpy(), None if want is None else want.float().cpu().numpy()))      # (by value: see tests/test_dist_gloo.py)
    _dist.barrier()
    _dist.destroy_process_group()


def test_sharded_path_through_rccl_with_one_rank(hip):
    """The collective calls of dist.upscale_sharded (all-reduce of the share sizes and of the frame-owner table, all-gather of
    the padded bf16 frame shares) issued through RCCL on device tensors -- in a one-rank "nccl" group, which is what a 1-GPU box
    can run; the world-2 test below needs a multi-GPU node.  Result == the plain single-GPU pipeline, bit for bit."""
    import socket
    import torch.distributed as tdist
    config, weights, dit, vae, runner, pipeline, d = (sub(n) for n in ("config", "weights", "dit", "vae", "runner", "pipeline", "dist"))
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        dcfg, vcfg = config.DIT_TINY, config.VAEConfig(block_out_channels=(128, 128, 128, 128))
        r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg))
        r.dit = dit.NaDiTEngine(dcfg, weights.synth_dit_state_dict(dcfg, seed=21), hip)
        r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=22), hip)
        images = torch.rand(13, 24, 40, 3, generator=torch.Generator().manual_seed(4)).cuda()
        txt = weights.synth_text_embedding().cuda()
        kw = dict(resolution=48, batch_size=5, uniform_batch_size=True, temporal_overlap=2, color_correction="lab")
        want = pipeline.upscale(images, r, txt, **kw)
        got = d.upscale_sharded(images, r, txt, force_collectives=True, **kw)
        assert torch.equal(got, want)
    finally:
        tdist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs on one node (RCCL over xGMI)")
def test_two_gpu_rccl_sharded_pipeline_equals_single_gpu():
    """One process per GPU, torch.distributed backend "nccl" (= RCCL): the sharded pipeline (round-robin temporal batches,
    point-to-point overlap heads, one all-gather of the upscaled frames) returns on every rank exactly what one GPU
    computes alone -- every kernel on the path is deterministic, so the comparison is bit-exact."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (o, w) for r, o, w in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    import numpy as np
    want = res[0][1]
    assert np.array_equal(res[0][0], want) and np.array_equal(res[1][0], want)


# ------------------------------------------------------------------ colour correction on the device (§8f N3: all five modes)
@pytest.mark.gpu
@pytest.mark.parametrize("method", ["adain", "wavelet", "lab", "hsv", "wavelet_adaptive"])
def test_colour_modes_on_the_device_equal_their_cpu_run(method):
    """The pipeline runs colour correction on the device the decoder left the frames on; tests/test_glue.py pins the same
    functions to the reference's text on CPU (<= 2e-6).  Here: device result == CPU result.  The three histogram modes
    are RANK based (sort + gather), so pixels tied within fp32 rounding of the device's blur / colour-space arithmetic
    may exchange matched values: all but a handful agree to 2e-5, none is off by more than a histogram neighbour.
    No exactly-tied pixels here (the CPU test's grey block): torch.sort is not stable, so the order of ties -- and with it
    which tied pixel receives which matched value -- is undefined between devices for the reference's own code as well."""
    cf = sub("colorfix")
    g = torch.Generator().manual_seed(11)
    content = (torch.rand(3, 3, 96, 160, generator=g) * 2 - 1) * 0.9
    style = (torch.rand(3, 3, 96, 160, generator=g) * 2 - 1) * 0.6 + 0.1
    want = cf.METHODS[method](content.clone(), style.clone())
    got = cf.METHODS[method](content.cuda(), style.cuda())
    assert got.is_cuda and got.shape == want.shape and got.dtype == want.dtype
    d = (got.cpu() - want).abs()
    assert torch.isfinite(got).all()
    if method in ("adain", "wavelet"):
        assert float(d.max()) < 2e-5, float(d.max())
    else:
        assert float((d > 2e-5).float().mean()) < 5e-3 and float(d.max()) < 5e-3, (float((d > 2e-5).float().mean()), float(d.max()))
