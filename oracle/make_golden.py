"""Generate the committed golden fixtures in tests/golden/ by running the REFERENCE implementation
(imported unmodified from /root/reference through oracle/third_party_shims.py) in fp32 on
seeded synthetic weights and inputs.  Run from the repo root in the build container:

    python -m oracle.make_golden [--skip-3b]

Fixtures (all tensors small; weights are NOT stored -- they are regenerated from the seed by
``<package>.weights`` on whichever machine runs the tests):
  text_pos_emb.pt     the reference's shipped prompt embedding (pos_emb.pt, [58,5120] bf16), data only
  dit_tiny.pt         DIT_TINY  forward, latent 3x24x40          (regular + shifted ragged windows)
  dit7b_tiny.pt       DIT_7B_TINY forward (the dit_7b model code), latent 3x24x40
  dit3b_cfg1.pt       full SeedVR2-3B forward, latent 1x32x32    (BASELINE config 1 shape)
  vae_small.pt        full VAE encode/decode of a 5x64x96 clip, untiled and tiled (32x48 / 16)
Round-2 fixtures at production width / headline geometry (``--only r2`` regenerates just these; the inputs are NOT
stored -- tests rebuild them with the seeded helpers below, which use only exact elementwise arithmetic):
  dit3b_w4l_crop.pt   SeedVR2-3B WIDTH (2560, 20 heads), 4 layers (2 MM + 2 shared, last vid-only) on a cropped
                      BASELINE config-3 grid: latent 9x60x108 -> 9x30x54 tokens; windows 3x15x27 = the config-3
                      window, 12 regular + 36 shifted windows with ragged edges (77 .. 1215 video rows)
  vae_tiled17.pt      full VAE, 17 frames 96x160, tiled 64/32 px: 2x4 tiles + the skipped-tile rule on both axes,
                      5 temporal slices through the reference's causal memory (split 4)
  vae_tile1024.pt     full VAE at the REAL tile size: 5 frames 1024x1152 (latent 2x128x144), tiled 1024/128 ->
                      a 2-tile strip with the 128-px cosine blend; encode stored whole, decode as 8 crops
  pipeline_small.pt   the WHOLE chain (``--only r2-pipe``): 11 frames 24x40 -> 48x80, batches of 5 with uniform padding
                      and a 2-frame overlap blend, LAB colour fix; the reference's own NaDiT (DIT_TINY width) and VAE
                      (4 x 128 channels) classes plus the reference's text of pad_video_temporal / SideResize /
                      DivisiblePad / blend_overlapping_frames / lab_color_transfer, driven by oracle/pipeline_oracle.py
Round-3 fixtures (``--only r3-dit32`` / ``r3-dit7b`` / ``r3-refbf16``):
  dit3b_32l_crop.pt   the FULL-DEPTH SeedVR2-3B (32 layers, 2560 wide) on the same cropped config-3 grid 9x30x54 tokens
                      (48 ragged windows per layer pair); fp32 reference output
  dit7b_w2l_crop.pt   SeedVR2-7B at PRODUCTION WIDTH (3072, 24 heads, 60 rotated dims, biased GELU MLP of 12288), 2 layers,
                      on a 5x30x54-token crop (regular + shifted windows); fp32 reference output
  refbf16.pt          what the REFERENCE ITSELF produces in bf16 (model.to(bfloat16), bf16 inputs: its production dtype) on the
                      inputs of dit3b_w4l_crop / vae_tiled17 / pipeline_small -- the yardstick "engine error <= reference-bf16
                      error" of tests/test_gpu_parity.py (stored as bf16 tensors)
"""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
PKG = "comfyui-seedvr2_videoupscaler_amd"


def _bf16_values(t):
    return t.to(torch.bfloat16)


def dit_inputs(T, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(T, H, W, 16, generator=g)
    latent = torch.randn(T, H, W, 16, generator=g) * 0.9152
    cond = torch.cat([latent, torch.ones(T, H, W, 1)], dim=-1)      # infer.py:54-78, task "sr"
    return _bf16_values(torch.cat([noise, cond], dim=-1))


def dit_r2_config(config):
    """3B width, 4 layers: blocks 0-1 separate vid/txt weights, 2-3 shared, block 3 video-only MLP."""
    return config.DiTConfig(num_layers=4, mm_layers=2)


def blocky_frames(T, H, W, seed, cell=16):
    """Frames in [-1, 1] with large-scale structure (nearest-upsampled low-res noise) plus fine noise; only exact
    elementwise ops, so the bf16 values are identical on every machine."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(3, T, (H + cell - 1) // cell, (W + cell - 1) // cell, generator=g) * 1.6 - 0.8
    x = lo.repeat_interleave(cell, dim=2).repeat_interleave(cell, dim=3)[:, :, :H, :W]
    x = x + (torch.rand(3, T, H, W, generator=g) - 0.5) * 0.25
    return _bf16_values(x.clamp(-1, 1))[None]                       # [1, 3, T, H, W]


def latent_input(T, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return _bf16_values(torch.randn(1, 16, T, H, W, generator=g))


CROPS_1024 = [(0, 0), (0, 1056), (928, 0), (464, 832), (464, 960), (100, 896), (928, 1056), (512, 300)]   # (y, x) of 96x96 crops; x 896..1024 is the blend seam


def run_reference_dit(rl, cfg, sd, vid, txt, dtype=torch.float32):
    """``dtype`` bfloat16: the reference's production regime (weights, activations and inputs in bf16)."""
    ref = rl.build_reference_dit(cfg.as_dict(), {k: v.to(dtype) for k, v in sd.items()})
    T, H, W, C = vid.shape
    with torch.no_grad():
        out = ref(vid=vid.to(dtype).reshape(-1, C), txt=txt.to(dtype),
                  vid_shape=torch.tensor([[T, H, W]]), txt_shape=torch.tensor([[txt.shape[0]]]),
                  timestep=torch.tensor([1000.0])).vid_sample
    return out.reshape(T, H, W, -1).contiguous()


def dit7b_r3_config(config):
    """7B production width, 2 layers (regular + shifted windows)."""
    import dataclasses
    return dataclasses.replace(config.DIT_7B, num_layers=2, mm_layers=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-3b", action="store_true")
    ap.add_argument("--only", default="", help="'r2': only the round-2 fixtures; 'r2-dit' / 'r2-vae17' / 'r2-vae1024': one of them; "
                                               "'r3-dit32' / 'r3-dit7b' / 'r3-refbf16': the round-3 fixtures")
    args = ap.parse_args()
    if args.only.startswith("r6"):
        return main_r6(args.only)
    if args.only.startswith("r4"):
        return main_r4(args.only)
    if args.only.startswith("r3"):
        return main_r3(args.only)
    if args.only:
        return main_r2(args.only)
    from oracle import reference_loader as rl
    assert rl.available(), "needs /root/reference"
    config = importlib.import_module(PKG + ".config")
    weights = importlib.import_module(PKG + ".weights")
    os.makedirs(GOLD, exist_ok=True)

    txt = torch.load(os.path.join(rl.REFERENCE_ROOT, "pos_emb.pt"), weights_only=True)
    torch.save(txt.clone(), os.path.join(GOLD, "text_pos_emb.pt"))

    # ---- DiT tiny
    cfg = config.DIT_TINY
    sd = weights.synth_dit_state_dict(cfg)
    vid = dit_inputs(3, 24, 40, seed=42)
    out = run_reference_dit(rl, cfg, sd, vid, txt)
    torch.save({"vid": vid, "out": out, "seed_weights": weights.SEED_WEIGHTS, "config": "DIT_TINY"},
               os.path.join(GOLD, "dit_tiny.pt"))
    print("dit_tiny", tuple(out.shape), float(out.std()))

    # ---- DiT 7B family (src/models/dit_7b), reduced width
    cfg = config.DIT_7B_TINY
    sd = weights.synth_dit_state_dict(cfg)
    out = run_reference_dit(rl, cfg, sd, vid, txt)
    torch.save({"vid": vid, "out": out, "seed_weights": weights.SEED_WEIGHTS, "config": "DIT_7B_TINY"},
               os.path.join(GOLD, "dit7b_tiny.pt"))
    print("dit7b_tiny", tuple(out.shape), float(out.std()))

    # ---- DiT 3B, BASELINE config 1
    if not args.skip_3b:
        cfg = config.DIT_3B
        t0 = time.time()
        sd = weights.synth_dit_state_dict(cfg)
        print("3B weights %.0fs" % (time.time() - t0))
        vid = dit_inputs(1, 32, 32, seed=42)
        t0 = time.time()
        out = run_reference_dit(rl, cfg, sd, vid, txt)
        print("3B reference fp32 forward %.0fs" % (time.time() - t0))
        torch.save({"vid": vid, "out": out, "seed_weights": weights.SEED_WEIGHTS, "config": "DIT_3B"},
                   os.path.join(GOLD, "dit3b_cfg1.pt"))
        print("dit3b_cfg1", tuple(out.shape), float(out.std()))
        del sd

    # ---- VAE
    vcfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(vcfg)
    ref = rl.build_reference_vae({k: v.float() for k, v in sd.items()})
    g = torch.Generator().manual_seed(43)
    # smooth-ish frames in [-1, 1]: low-res noise, bicubic up (mimics an upscaled frame)
    lo = torch.rand(1, 3 * 5, 16, 24, generator=g) * 2 - 1
    x = torch.nn.functional.interpolate(lo, size=(64, 96), mode="bicubic", align_corners=False)
    x = x.clamp(-1, 1).reshape(1, 5, 3, 64, 96).permute(0, 2, 1, 3, 4).contiguous()
    x = _bf16_values(x)
    z_in = _bf16_values(torch.randn(1, 16, 2, 8, 12, generator=g))
    tile = dict(tiled=True, tile_size=(32, 48), tile_overlap=(16, 16))
    with torch.no_grad():
        enc = ref.encode(x.float()).latent
        enc_t = ref.encode(x.float(), **tile).latent
        dec = ref.decode(z_in.float()).sample
        dec_t = ref.decode(z_in.float(), **tile).sample
    torch.save({"x": x, "z_in": z_in, "enc": enc, "enc_tiled": enc_t, "dec": dec, "dec_tiled": dec_t,
                "tile_size": (32, 48), "tile_overlap": (16, 16), "seed_weights": weights.SEED_WEIGHTS + 1},
               os.path.join(GOLD, "vae_small.pt"))
    print("vae_small enc", tuple(enc.shape), "dec", tuple(dec.shape))


PIPE_CASE = dict(frames=11, hw=(24, 40), resolution=48, batch_size=5, temporal_overlap=2, uniform_batch_size=True,
                 seed_images=4, seed_dit=21, seed_vae=22, vae_channels=(128, 128, 128, 128))


def pipeline_noise(lat):
    """The (base, extra) noise pair of a batch as a function of the latent's size (what tests/test_gpu_parity.py injects)."""
    gg = torch.Generator().manual_seed(lat.numel())
    return torch.randn(lat.shape, generator=gg), torch.randn(lat.shape, generator=gg)


# Round 4: the same chain at PRODUCTION width and depth -- the reference's 32-layer SeedVR2-3B NaDiT and its full-width VAE
# (128, 256, 512, 512), VAE tiled 64 / 16 px in encode and decode (per-tile GroupNorm statistics and attention, cosine seams).
PIPE_PROD = dict(frames=9, hw=(48, 80), resolution=96, batch_size=5, temporal_overlap=2, uniform_batch_size=True,
                 seed_images=5, seed_dit=1234, seed_vae=1235, vae_channels=(128, 256, 512, 512), dit="DIT_3B",
                 vae_tile=(64, 64), vae_tile_overlap=(16, 16))


def reference_pipeline_components(rl, config, weights, txt, dtype=torch.float32, case=None):
    """pipeline_oracle.Components built from the REFERENCE: its NaDiT and VAE classes, and its glue function text.
    ``dtype`` bfloat16: models and the tensors handed to them in bf16 (the reference's production regime)."""
    from oracle import pipeline_oracle as po
    case = PIPE_CASE if case is None else case
    glue = rl.reference_glue()
    dcfg = getattr(config, case.get("dit", "DIT_TINY"))
    vcfg = config.VAEConfig(block_out_channels=case["vae_channels"])
    dsd = weights.synth_dit_state_dict(dcfg, seed=case["seed_dit"])
    for k in list(dsd):                                 # (in place: the 3B state dict is 13.6 GB in fp32)
        dsd[k] = dsd[k].to(dtype)
    vsd = weights.synth_vae_state_dict(vcfg, seed=case["seed_vae"])
    dit = rl.build_reference_dit(dcfg.as_dict(), dsd)
    vae = rl.build_reference_vae({k: v.to(dtype) for k, v in vsd.items()}, block_out_channels=vcfg.block_out_channels)
    tile = (dict(tiled=True, tile_size=tuple(case["vae_tile"]), tile_overlap=tuple(case["vae_tile_overlap"]))
            if case.get("vae_tile") else {})
    res = case["resolution"]
    resize, pad16 = glue["SideResize"](size=res, max_size=0), glue["DivisiblePad"]((16, 16))

    class _Dbg:
        def log(self, *a, **k):
            pass

    def video_transform(x_tchw):                       # generation_utils.py:72-84
        y = pad16(torch.clamp(resize(x_tchw), 0.0, 1.0))
        return ((y - 0.5) / 0.5).permute(1, 0, 2, 3)

    def true_dims(h, w):                               # generation_utils.py:124-137
        y = resize(torch.zeros(1, 3, h, w))
        return (y.shape[-2] // 2) * 2, (y.shape[-1] // 2) * 2

    def vae_encode(x_cthw):                            # infer.py:117-199
        with torch.no_grad():
            lat = vae.encode(x_cthw[None].to(dtype), **tile).latent
        lat = (lat.unsqueeze(2) if lat.ndim == 4 else lat)[0]          # infer.py:187 (a one-frame clip comes back as an image)
        return (lat.permute(1, 2, 3, 0) - vcfg.shifting_factor) * vcfg.scaling_factor

    def vae_decode(lat_thwc):                          # infer.py:203-278
        z = (lat_thwc / vcfg.scaling_factor + vcfg.shifting_factor).permute(3, 0, 1, 2)[None]
        with torch.no_grad():
            return vae.decode(z.to(dtype), **tile).sample[0].float()

    def dit_fn(vid, text):
        T, H, W, C = vid.shape
        with torch.no_grad():
            out = dit(vid=vid.to(dtype).reshape(-1, C), txt=text.to(dtype), vid_shape=torch.tensor([[T, H, W]]),
                      txt_shape=torch.tensor([[text.shape[0]]]), timestep=torch.tensor([1000.0])).vid_sample
        return out.reshape(T, H, W, -1)

    return po.Components(
        pad_video_temporal=glue["pad_video_temporal"], video_transform=video_transform, true_target_dims=true_dims,
        vae_encode=vae_encode, dit=dit_fn, vae_decode=vae_decode, blend_overlapping_frames=glue["blend_overlapping_frames"],
        color_fix=lambda s_, r_: glue["lab_color_transfer"](s_, r_, _Dbg(), luminance_weight=0.8),
        noise=pipeline_noise if dtype == torch.float32 else (lambda lat: tuple(n.to(lat.dtype) for n in pipeline_noise(lat))))


def main_r2(which):
    from oracle import reference_loader as rl
    assert rl.available(), "needs /root/reference"
    config = importlib.import_module(PKG + ".config")
    weights = importlib.import_module(PKG + ".weights")
    txt = torch.load(os.path.join(GOLD, "text_pos_emb.pt"), weights_only=True)
    if which in ("r2", "r2-pipe"):
        from oracle import pipeline_oracle as po
        pc = PIPE_CASE
        images = torch.rand(pc["frames"], pc["hw"][0], pc["hw"][1], 3, generator=torch.Generator().manual_seed(pc["seed_images"]))
        text = weights.synth_text_embedding()
        comps = reference_pipeline_components(rl, config, weights, text)
        t0 = time.time()
        out = po.upscale(images, text.float(), comps, pc["batch_size"], pc["temporal_overlap"], pc["uniform_batch_size"])
        print("pipeline_small %.0fs" % (time.time() - t0), tuple(out.shape), float(out.mean()), float(out.std()))
        torch.save({"out": out.clone(), **{k: v for k, v in pc.items()}}, os.path.join(GOLD, "pipeline_small.pt"))
    if which in ("r2", "r2-dit"):
        cfg = dit_r2_config(config)
        sd = weights.synth_dit_state_dict(cfg)
        vid = dit_inputs(9, 60, 108, seed=44)
        t0 = time.time()
        out = run_reference_dit(rl, cfg, sd, vid, txt)
        print("dit3b_w4l_crop reference fp32 forward %.0fs" % (time.time() - t0), tuple(out.shape), float(out.std()))
        torch.save({"out": out, "latent": (9, 60, 108), "seed_input": 44, "seed_weights": weights.SEED_WEIGHTS},
                   os.path.join(GOLD, "dit3b_w4l_crop.pt"))
        del sd
    if which in ("r2", "r2-vae17", "r2-vae1024"):
        vcfg = config.VAE_V3
        sd = weights.synth_vae_state_dict(vcfg)
        ref = rl.build_reference_vae({k: v.float() for k, v in sd.items()})
    if which in ("r2", "r2-vae17"):
        x = blocky_frames(17, 96, 160, seed=45, cell=8)
        z = latent_input(5, 12, 20, seed=46)
        tile = dict(tiled=True, tile_size=(64, 64), tile_overlap=(32, 32))
        t0 = time.time()
        with torch.no_grad():
            enc = ref.encode(x.float(), **tile).latent
            dec = ref.decode(z.float(), **tile).sample
        print("vae_tiled17 %.0fs" % (time.time() - t0), tuple(enc.shape), tuple(dec.shape))
        torch.save({"enc_tiled": enc.clone(), "dec_tiled": dec.clone(), "frames": (17, 96, 160), "latent": (5, 12, 20), "seed_x": 45,
                    "seed_z": 46, "cell": 8, "tile_size": (64, 64), "tile_overlap": (32, 32),
                    "seed_weights": weights.SEED_WEIGHTS + 1}, os.path.join(GOLD, "vae_tiled17.pt"))
    if which in ("r2", "r2-vae1024"):
        x = blocky_frames(5, 1024, 1152, seed=47, cell=32)
        z = latent_input(2, 128, 144, seed=48)
        tile = dict(tiled=True, tile_size=(1024, 1024), tile_overlap=(128, 128))
        t0 = time.time()
        with torch.no_grad():
            enc = ref.encode(x.float(), **tile).latent
        print("vae_tile1024 encode %.0fs" % (time.time() - t0), tuple(enc.shape))
        t0 = time.time()
        with torch.no_grad():
            dec = ref.decode(z.float(), **tile).sample
        print("vae_tile1024 decode %.0fs" % (time.time() - t0), tuple(dec.shape))
        crops = torch.stack([dec[0, :, :, y:y + 96, xx:xx + 96] for (y, xx) in CROPS_1024])
        torch.save({"enc_tiled": enc.clone(), "dec_crops": crops, "crops": CROPS_1024, "dec_mean": float(dec.mean()),
                    "dec_std": float(dec.std()), "dec_min": float(dec.min()), "dec_max": float(dec.max()),
                    "frames": (5, 1024, 1152), "latent": (2, 128, 144), "seed_x": 47, "seed_z": 48, "cell": 32,
                    "tile_size": (1024, 1024), "tile_overlap": (128, 128), "seed_weights": weights.SEED_WEIGHTS + 1},
                   os.path.join(GOLD, "vae_tile1024.pt"))


def main_r3(which):
    from oracle import reference_loader as rl
    assert rl.available(), "needs /root/reference"
    config = importlib.import_module(PKG + ".config")
    weights = importlib.import_module(PKG + ".weights")
    txt = torch.load(os.path.join(GOLD, "text_pos_emb.pt"), weights_only=True)
    if which in ("r3", "r3-dit32"):
        cfg = config.DIT_3B
        sd = weights.synth_dit_state_dict(cfg)
        vid = dit_inputs(9, 60, 108, seed=44)
        t0 = time.time()
        out = run_reference_dit(rl, cfg, sd, vid, txt)
        print("dit3b_32l_crop reference fp32 forward %.0fs" % (time.time() - t0), tuple(out.shape), float(out.std()))
        torch.save({"out": out, "latent": (9, 60, 108), "seed_input": 44, "seed_weights": weights.SEED_WEIGHTS},
                   os.path.join(GOLD, "dit3b_32l_crop.pt"))
        del sd
    if which in ("r3", "r3-dit7b"):
        cfg = dit7b_r3_config(config)
        sd = weights.synth_dit_state_dict(cfg)
        vid = dit_inputs(5, 60, 108, seed=49)
        t0 = time.time()
        out = run_reference_dit(rl, cfg, sd, vid, txt)
        print("dit7b_w2l_crop reference fp32 forward %.0fs" % (time.time() - t0), tuple(out.shape), float(out.std()))
        torch.save({"out": out, "latent": (5, 60, 108), "seed_input": 49, "seed_weights": weights.SEED_WEIGHTS},
                   os.path.join(GOLD, "dit7b_w2l_crop.pt"))
        del sd
    if which in ("r3", "r3-refbf16"):
        bf = torch.bfloat16
        res = {}
        # DiT at production width (the inputs of dit3b_w4l_crop)
        cfg = dit_r2_config(config)
        out = run_reference_dit(rl, cfg, weights.synth_dit_state_dict(cfg), dit_inputs(9, 60, 108, seed=44), txt, dtype=bf)
        res["dit3b_w4l_crop"] = out.to(bf)
        print("refbf16 dit3b_w4l_crop", tuple(out.shape))
        # VAE (the inputs of vae_tiled17)
        vcfg = config.VAE_V3
        ref = rl.build_reference_vae({k: v.to(bf) for k, v in weights.synth_vae_state_dict(vcfg).items()})
        tile = dict(tiled=True, tile_size=(64, 64), tile_overlap=(32, 32))
        t0 = time.time()
        with torch.no_grad():
            res["vae_tiled17_enc"] = ref.encode(blocky_frames(17, 96, 160, seed=45, cell=8).to(bf), **tile).latent.to(bf)
            res["vae_tiled17_dec"] = ref.decode(latent_input(5, 12, 20, seed=46).to(bf), **tile).sample.to(bf)
        print("refbf16 vae_tiled17 %.0fs" % (time.time() - t0))
        del ref
        # the whole chain (the inputs of pipeline_small)
        from oracle import pipeline_oracle as po
        pc = PIPE_CASE
        images = torch.rand(pc["frames"], pc["hw"][0], pc["hw"][1], 3, generator=torch.Generator().manual_seed(pc["seed_images"]))
        text = weights.synth_text_embedding()
        comps = reference_pipeline_components(rl, config, weights, text, dtype=bf)
        out = po.upscale(images, text, comps, pc["batch_size"], pc["temporal_overlap"], pc["uniform_batch_size"], compute_dtype=bf)
        res["pipeline_small"] = out.to(bf)
        print("refbf16 pipeline_small", tuple(out.shape))
        torch.save(res, os.path.join(GOLD, "refbf16.pt"))


def main_r4(which):
    from oracle import reference_loader as rl
    assert rl.available(), "needs /root/reference"
    config = importlib.import_module(PKG + ".config")
    weights = importlib.import_module(PKG + ".weights")
    if which in ("r4", "r4-prod"):
        from oracle import pipeline_oracle as po
        pc = PIPE_PROD
        images = torch.rand(pc["frames"], pc["hw"][0], pc["hw"][1], 3, generator=torch.Generator().manual_seed(pc["seed_images"]))
        text = weights.synth_text_embedding()
        t0 = time.time()
        comps = reference_pipeline_components(rl, config, weights, text, case=pc)
        print("pipeline_prod reference models built %.0fs" % (time.time() - t0))
        t0 = time.time()
        out = po.upscale(images, text.float(), comps, pc["batch_size"], pc["temporal_overlap"], pc["uniform_batch_size"])
        print("pipeline_prod fp32 %.0fs" % (time.time() - t0), tuple(out.shape), float(out.mean()), float(out.std()))
        del comps
        t0 = time.time()
        comps = reference_pipeline_components(rl, config, weights, text, dtype=torch.bfloat16, case=pc)
        out_bf = po.upscale(images, text, comps, pc["batch_size"], pc["temporal_overlap"], pc["uniform_batch_size"],
                            compute_dtype=torch.bfloat16)
        mse = float((out_bf.double() - out.double()).pow(2).mean())
        import math
        print("pipeline_prod reference-bf16 twin %.0fs: %.2f dB vs its fp32 self" % (time.time() - t0, 10 * math.log10(1.0 / mse)))
        del comps
        torch.save({"out": out.clone(), "out_refbf16": out_bf.to(torch.bfloat16).clone(), **{k: v for k, v in pc.items()}},
                   os.path.join(GOLD, "pipeline_prod.pt"))
    if which in ("r4", "r4-dit7b36"):
        # the FULL-DEPTH SeedVR2-7B (36 layers, 3072 wide, 8.2e9 parameters = 33 GB in fp32) at BASELINE config 1's shape
        txt = torch.load(os.path.join(GOLD, "text_pos_emb.pt"), weights_only=True)
        cfg = config.DIT_7B
        t0 = time.time()
        sd = weights.synth_dit_state_dict(cfg)
        for k in list(sd):
            sd[k] = sd[k].float()
        print("7B weights %.0fs" % (time.time() - t0))
        vid = dit_inputs(1, 32, 32, seed=50)
        ref = rl.build_reference_dit(cfg.as_dict(), sd)
        del sd
        t0 = time.time()
        with torch.no_grad():
            out = ref(vid=vid.float().reshape(-1, 33), txt=txt.float(), vid_shape=torch.tensor([[1, 32, 32]]),
                      txt_shape=torch.tensor([[txt.shape[0]]]), timestep=torch.tensor([1000.0])).vid_sample
        out = out.reshape(1, 32, 32, -1).contiguous()
        print("dit7b_36l_cfg1 reference fp32 forward %.0fs" % (time.time() - t0), tuple(out.shape), float(out.std()))
        torch.save({"out": out, "latent": (1, 32, 32), "seed_input": 50, "seed_weights": weights.SEED_WEIGHTS},
                   os.path.join(GOLD, "dit7b_36l_cfg1.pt"))


# Round 6: BASELINE's own configurations end to end against the reference.
# config 1 IN FULL: one image -> 256 x 256 (latent 1 x 32 x 32 -> 256 video + 58 text tokens), the reference's 32-layer SeedVR2-3B
# NaDiT + full-width VAE, untiled, no temporal batching; same model seeds as PIPE_PROD so that the GPU tests share one 3B engine.
PIPE_CFG1 = dict(frames=1, hw=(128, 128), resolution=256, batch_size=1, temporal_overlap=0, uniform_batch_size=False,
                 seed_images=6, seed_dit=1234, seed_vae=1235, vae_channels=(128, 256, 512, 512), dit="DIT_3B")
# config 2's GEOMETRY, untiled (attn_video_vae.py:1254-1300 slicing, :615-665 per-frame attention, causal_inflation_lib.py:354-409
# per-frame GroupNorm): one 2048 x 2048 frame = GroupNorm groups over 4.2e6 pixels and 65 536-token mid-block attention, plus a
# 5-frame 1024 x 1024 clip (two temporal slices through the causal memory, 16 384-token attention).  The 5 x 2048 x 2048 clip the
# review named first needs ~10.7 GB per full-resolution fp32 activation of the reference on the 62 GB build box: not run.
VAE_BIG = {"vae_untiled_2048": dict(frames=(1, 2048, 2048), latent=(1, 256, 256), seed_x=51, seed_z=52, cell=32),
           "vae_untiled_1024x5": dict(frames=(5, 1024, 1024), latent=(2, 128, 128), seed_x=53, seed_z=54, cell=32)}


def big_crops(H, W, size=96):
    """(y, x) of 96 x 96 crops of a decoded frame: the four corners, the centre, three interior points."""
    return [(0, 0), (0, W - size), (H - size, 0), (H - size, W - size), (H // 2 - size // 2, W // 2 - size // 2),
            (H // 4, W // 3), (H // 3 * 2, W // 5), (H // 5, W // 3 * 2)]


# Round 6: heavy-tailed statistics.  Every other fixture draws its weights N(0, 1 / fan_in) with gamma = 1, beta = 0 -- benign next to a
# trained checkpoint.  These two modifiers (deterministic; tests rebuild the same state dicts) give the h16 storage regime (range
# +-4.2e6, absolute floor 3.8e-6) and its overflow guards what a real checkpoint can hold: "massive activation" channels in the NaDiT's
# residual stream, modulation scales and GroupNorm gains spread over three decades, biases of order one, one conv with 10x weights,
# and every matrix rounded to e4m3 values first (a fp8 checkpoint as the reference up-casts it, compatibility.py:895-938).
# ``level`` "tail": everything stays finite in h16 (the guards must stay silent);  "overflow": one producer is scaled so that the
# residual stream / trunk leaves h16's range while bf16 / fp32 -- what the reference runs in -- hold it easily (the guard must re-run
# the call with fp32 stores ON THE HIP PATH and still match).
HEAVY_DIT = dict(latent=(3, 24, 40), seed_input=61, hot_channels=(7, 100, 201), boost={"tail": 300.0, "overflow": 3.0e7})
HEAVY_VAE = dict(frames=(5, 32, 48), latent=(2, 4, 6), seed_x=62, seed_z=63, cell=8,
                 loud_conv="decoder.up_blocks.2.resnets.1.conv1.weight", loud=10.0,
                 overflow_convs=("encoder.mid_block.resnets.0.conv2", "decoder.mid_block.resnets.0.conv2"), overflow_boost=3.0e5)


def _log_uniform(n, lo, hi, g):
    import math
    return torch.exp(torch.rand(n, generator=g) * (math.log(hi) - math.log(lo)) + math.log(lo))


def heavy_dit_state_dict(sd, level="tail"):
    g = torch.Generator().manual_seed(77)
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if v.dim() == 2 and v.dtype == torch.bfloat16:                       # e4m3-valued matrices, held in bf16 like an up-cast fp8 file
            v = v.float().clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.bfloat16)
        if k.endswith(("attn_scale", "mlp_scale", "out_scale")):           # modulation scales over three decades
            v = (v.float() * _log_uniform(v.numel(), 0.05, 8.0, g)).to(v.dtype)
        out[k] = v
    hot, boost = list(HEAVY_DIT["hot_channels"]), HEAVY_DIT["boost"][level]
    for name in ("vid_in.proj.weight", "vid_in.proj.bias", "txt_in.weight", "txt_in.bias"):
        w = out[name].float()
        w[hot] *= boost                                                      # residual-stream channels ~boost x the others
        out[name] = w.to(out[name].dtype)
    return out


def heavy_vae_state_dict(sd, level="tail"):
    g = torch.Generator().manual_seed(78)
    out = {k: v.clone() for k, v in sd.items()}
    for k in sorted(out):
        if ("norm" in k) and k.endswith(".weight") and out[k].dim() == 1:   # GroupNorm gains, log-uniform in [0.01, 30]
            out[k] = _log_uniform(out[k].numel(), 0.01, 30.0, g).to(out[k].dtype)
        elif ("norm" in k) and k.endswith(".bias"):
            out[k] = (torch.rand(out[k].numel(), generator=g) * 4 - 2).to(out[k].dtype)
    out[HEAVY_VAE["loud_conv"]] = (out[HEAVY_VAE["loud_conv"]].float() * HEAVY_VAE["loud"]).to(torch.bfloat16)
    if level == "overflow":
        for name in HEAVY_VAE["overflow_convs"]:
            for suffix in (".weight", ".bias"):
                out[name + suffix] = (out[name + suffix].float() * HEAVY_VAE["overflow_boost"]).to(torch.bfloat16)
    return out


def main_r6_heavy():
    from oracle import reference_loader as rl
    assert rl.available(), "needs /root/reference"
    config = importlib.import_module(PKG + ".config")
    weights = importlib.import_module(PKG + ".weights")
    bf = torch.bfloat16
    txt = torch.load(os.path.join(GOLD, "text_pos_emb.pt"), weights_only=True)
    res = {"dit": dict(HEAVY_DIT), "vae": dict(HEAVY_VAE)}
    cfg = config.DIT_TINY
    vid = dit_inputs(*HEAVY_DIT["latent"], seed=HEAVY_DIT["seed_input"])
    for level in ("tail", "overflow"):
        sd = heavy_dit_state_dict(weights.synth_dit_state_dict(cfg), level)
        out = run_reference_dit(rl, cfg, sd, vid, txt)
        out_bf = run_reference_dit(rl, cfg, sd, vid, txt, dtype=bf)
        e = float((out_bf.double() - out.double()).norm() / out.double().norm())
        print(f"dit_tiny_heavy[{level}] out std {float(out.std()):.3g}, max {float(out.abs().max()):.3g}; reference bf16 vs fp32 rel-err {e:.3e}")
        res[f"dit_{level}"] = out.clone()
        res[f"dit_{level}_refbf16"] = out_bf.to(bf).clone()
    vcfg = config.VAE_V3
    x = blocky_frames(*HEAVY_VAE["frames"], seed=HEAVY_VAE["seed_x"], cell=HEAVY_VAE["cell"])
    z = latent_input(*HEAVY_VAE["latent"], seed=HEAVY_VAE["seed_z"])
    for level in ("tail", "overflow"):
        sd = heavy_vae_state_dict(weights.synth_vae_state_dict(vcfg), level)
        for dt, tag in ((torch.float32, ""), (bf, "_refbf16")):
            ref = rl.build_reference_vae({k: v.to(dt) for k, v in sd.items()})
            with torch.no_grad():
                enc = ref.encode(x.to(dt)).latent
                dec = ref.decode(z.to(dt)).sample
            res[f"vae_{level}_enc{tag}"] = enc.to(dt).clone()
            res[f"vae_{level}_dec{tag}"] = dec.to(dt).clone()
        e_enc = float((res[f"vae_{level}_enc_refbf16"].double() - res[f"vae_{level}_enc"].double()).norm() / res[f"vae_{level}_enc"].double().norm())
        e_dec = float((res[f"vae_{level}_dec_refbf16"].double() - res[f"vae_{level}_dec"].double()).norm() / res[f"vae_{level}_dec"].double().norm())
        print(f"vae_heavy[{level}] enc std {float(res[f'vae_{level}_enc'].std()):.3g} dec std {float(res[f'vae_{level}_dec'].std()):.3g} "
              f"max {float(res[f'vae_{level}_dec'].abs().max()):.3g}; reference bf16 vs fp32 rel-err enc {e_enc:.3e} dec {e_dec:.3e}")
    torch.save(res, os.path.join(GOLD, "heavy_tail.pt"))


# Round 6: the sampler beyond the pipeline's forced steps = 1 / cfg = 1 (generation_phases.py:599-601): the reference's own EulerSampler
# over several trailing timesteps with classifier-free guidance (partial + rescale), wired exactly as VideoDiffusionInfer.inference wires
# it (infer.py:315-395) -- what `runner.inference` refused with NotImplementedError until now.  Two clips of different sizes, each
# sampled as a batch of ONE: the reference's batched NaDiT call is not a usable mode -- with more than one clip its window attention
# hands the clips' text to the windows round-robin (na.py:381-387 `batch_idx = i % batch_size`, `len(vid_len) // batch_size` repeats), so
# clip 0's output moves by 3.4 % when only clip 1's TEXT changes (measured with the imported reference, DIT_TINY) -- while
# `NaDiTEngine.__call__` / `runner.inference` treat a batch as what it means: independent clips.
SAMPLER_CASE = dict(latents=((3, 16, 24), (1, 12, 16)), steps=4, cfg_scale=2.5, cfg_partial=0.5, cfg_rescale=0.7, seed=71)


def sampler_inputs():
    """-> (noises, conditions, texts_pos, texts_neg): bf16-valued, rebuilt from the seed by tests."""
    g = torch.Generator().manual_seed(SAMPLER_CASE["seed"])
    noises, conds, tp, tn = [], [], [], []
    for (T, H, W) in SAMPLER_CASE["latents"]:
        noises.append(_bf16_values(torch.randn(T, H, W, 16, generator=g)))
        lat = torch.randn(T, H, W, 16, generator=g) * 0.9152
        conds.append(_bf16_values(torch.cat([lat, torch.ones(T, H, W, 1)], dim=-1)))
        tp.append(_bf16_values(torch.randn(58, 5120, generator=g) * 0.5))
        tn.append(_bf16_values(torch.randn(58, 5120, generator=g) * 0.5))
    return noises, conds, tp, tn


def main_r6_sampler():
    from oracle import reference_loader as rl
    assert rl.available(), "needs /root/reference"
    config = importlib.import_module(PKG + ".config")
    weights = importlib.import_module(PKG + ".weights")
    sc = SAMPLER_CASE
    cfg = config.DIT_TINY
    dit = rl.build_reference_dit(cfg.as_dict(), {k: v.float() for k, v in weights.synth_dit_state_dict(cfg).items()})
    st = rl.reference_sampler_stack(T=1000.0, steps=sc["steps"])
    na, sampler, cfg_dispatch = st["na"], st["sampler"], st["cfg"]
    noises, conds, tp, tn = (list(t.float() for t in ts) for ts in sampler_inputs())

    def sample(idx, guided):                               # infer.py:334-386 for the one-clip batch [idx]
        batch = 1
        pos_e, pos_s = na.flatten(tp[idx:idx + 1])
        neg_e, neg_s = na.flatten(tn[idx:idx + 1])
        latents, shapes = na.flatten(noises[idx:idx + 1])
        lat_cond, _ = na.flatten(conds[idx:idx + 1])
        with torch.no_grad():
            out = sampler.sample(
                x=latents,
                f=lambda args: cfg_dispatch(
                    pos=lambda: dit(vid=torch.cat([args.x_t, lat_cond], dim=-1), txt=pos_e, vid_shape=shapes, txt_shape=pos_s,
                                    timestep=args.t.repeat(batch)).vid_sample,
                    neg=lambda: dit(vid=torch.cat([args.x_t, lat_cond], dim=-1), txt=neg_e, vid_shape=shapes, txt_shape=neg_s,
                                    timestep=args.t.repeat(batch)).vid_sample,
                    scale=((sc["cfg_scale"] if guided else 1.0) if (args.i + 1) / len(sampler.timesteps) <= sc["cfg_partial"] else 1.0),
                    rescale=sc["cfg_rescale"]))
        return na.unflatten(out, shapes)[0]

    outs = [sample(0, True), sample(1, True)]
    plain = sample(0, False)                               # several steps alone, guidance off
    print("sampler_multistep", [tuple(o.shape) for o in outs], [float(o.std()) for o in outs], float(plain.std()))
    torch.save({"outs": [o.clone() for o in outs], "plain": plain.clone(),
                "timesteps": sampler.timesteps.timesteps.clone(), **sc}, os.path.join(GOLD, "sampler_multistep.pt"))


def main_r6(which):
    import math
    if which in ("r6", "r6-heavy"):
        main_r6_heavy()
        if which == "r6-heavy":
            return
    if which in ("r6", "r6-sampler"):
        main_r6_sampler()
        if which == "r6-sampler":
            return
    from oracle import reference_loader as rl
    assert rl.available(), "needs /root/reference"
    config = importlib.import_module(PKG + ".config")
    weights = importlib.import_module(PKG + ".weights")
    if which in ("r6", "r6-cfg1"):
        from oracle import pipeline_oracle as po
        pc = PIPE_CFG1
        images = torch.rand(pc["frames"], pc["hw"][0], pc["hw"][1], 3, generator=torch.Generator().manual_seed(pc["seed_images"]))
        text = weights.synth_text_embedding()
        comps = reference_pipeline_components(rl, config, weights, text, case=pc)
        t0 = time.time()
        out = po.upscale(images, text.float(), comps, pc["batch_size"], pc["temporal_overlap"], pc["uniform_batch_size"])
        print("pipeline_cfg1 fp32 %.1fs" % (time.time() - t0), tuple(out.shape), float(out.mean()), float(out.std()))
        del comps
        comps = reference_pipeline_components(rl, config, weights, text, dtype=torch.bfloat16, case=pc)
        out_bf = po.upscale(images, text, comps, pc["batch_size"], pc["temporal_overlap"], pc["uniform_batch_size"],
                            compute_dtype=torch.bfloat16)
        mse = float((out_bf.double() - out.double()).pow(2).mean())
        print("pipeline_cfg1 reference-bf16 twin: %.2f dB vs its fp32 self" % (10 * math.log10(1.0 / mse)))
        del comps
        torch.save({"out": out.clone(), "out_refbf16": out_bf.to(torch.bfloat16).clone(), **{k: v for k, v in pc.items()}},
                   os.path.join(GOLD, "pipeline_cfg1.pt"))
    for name, c in VAE_BIG.items():
        if which not in ("r6", "r6-vaebig", "r6-" + name):
            continue
        vcfg = config.VAE_V3
        sd = weights.synth_vae_state_dict(vcfg)
        ref = rl.build_reference_vae({k: v.float() for k, v in sd.items()})
        x = blocky_frames(*c["frames"], seed=c["seed_x"], cell=c["cell"])
        z = latent_input(*c["latent"], seed=c["seed_z"])
        t0 = time.time()
        with torch.no_grad():
            enc = ref.encode(x.float()).latent
        print(name, "encode %.0fs" % (time.time() - t0), tuple(enc.shape), flush=True)
        t0 = time.time()
        with torch.no_grad():
            dec = ref.decode(z.float()).sample
        print(name, "decode %.0fs" % (time.time() - t0), tuple(dec.shape), flush=True)
        if dec.dim() == 4:                                   # (a one-frame clip comes back as an image [1, 3, H, W])
            dec = dec.unsqueeze(2)
        if enc.dim() == 4:
            enc = enc.unsqueeze(2)
        crops = big_crops(dec.shape[-2], dec.shape[-1])
        torch.save({"enc": enc.clone(), "dec_crops": torch.stack([dec[0, :, :, y:y + 96, xx:xx + 96] for (y, xx) in crops]),
                    "crops": crops, "dec_mean": float(dec.mean()), "dec_std": float(dec.std()), **c,
                    "seed_weights": weights.SEED_WEIGHTS + 1}, os.path.join(GOLD, name + ".pt"))
        del ref, enc, dec


if __name__ == "__main__":
    main()
