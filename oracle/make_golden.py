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


def run_reference_dit(rl, cfg, sd, vid, txt):
    ref = rl.build_reference_dit(cfg.as_dict(), {k: v.float() for k, v in sd.items()})
    T, H, W, C = vid.shape
    with torch.no_grad():
        out = ref(vid=vid.float().reshape(-1, C), txt=txt.float(),
                  vid_shape=torch.tensor([[T, H, W]]), txt_shape=torch.tensor([[txt.shape[0]]]),
                  timestep=torch.tensor([1000.0])).vid_sample
    return out.reshape(T, H, W, -1).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-3b", action="store_true")
    args = ap.parse_args()
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


if __name__ == "__main__":
    main()
