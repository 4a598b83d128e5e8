"""-m gpu, round 4: the two places where the product path had only met the CPU double of the C ABI.

1. CHECKPOINT INGEST ON THE HIP PATH (reference: src/core/model_loader.py:84-153, 818-833,
   compatibility.py:895-938, causal_inflation_lib.py:440-503).  Checkpoint FILES in the registry's precisions (fp16 and
   fp8-e4m3 safetensors, ComfyUI key prefix, a VAE file holding 2D image-VAE kernels) go through
   ``checkpoint.build_engines(HipOps, ...)`` and the engines' outputs are checked against the reference's goldens / the CPU oracle.
2. THE DROP-IN EXECUTED ON THE PRODUCT PATH: the reference's own four phase functions (generation_phases.py:171, 542, 807, 1060,
   loaded by oracle/reference_loader.py from the checkout or from the byte-compiled oracle/_ref that travels to the GPU box)
   drive ``runner.VideoDiffusionInfer`` with HIP engines on cuda:0, exactly as they drive the reference's runner.
"""
import math
import os

import pytest
import torch

from conftest import sub, rel_err, GOLDEN
from oracle import reference_loader as rl

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def hip():
    return sub("ops").HipOps("cuda:0")


def _golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


def _psnr(a, b, peak):
    mse = float((a.double() - b.double()).pow(2).mean())
    return 10 * math.log10(peak * peak / max(mse, 1e-30))


# ------------------------------------------------------------------------------------------------------------------------
def test_dit_checkpoint_files_fp16_and_fp8_through_hipops(hip, tmp_path):
    """fp16 file -> the engine reproduces the reference golden made with the bf16 weights (fp16 holds every bf16 value that does
    not underflow); fp8-e4m3 file with the ComfyUI prefix -> every weight is bf16(e4m3 value), an exact up-cast, so the engine
    must match the CPU oracle run on exactly those values."""
    from safetensors.torch import save_file
    from oracle import dit_oracle
    ck, weights, config, windows = sub("checkpoint"), sub("weights"), sub("config"), sub("windows")
    g, txt = _golden("dit_tiny.pt"), _golden("text_pos_emb.pt")
    cfg = config.DIT_TINY
    sd = weights.synth_dit_state_dict(cfg, seed=g["seed_weights"])
    p16 = str(tmp_path / "seedvr2_ema_3b_fp16.safetensors")
    save_file({k: v.to(torch.float16).contiguous() for k, v in sd.items()}, p16)
    eng, _ = ck.build_engines(hip, dit_path=p16, dit_cfg=cfg)
    out = eng.forward(g["vid"].cuda(), txt.cuda(), 1000.0).float().cpu()
    e = rel_err(out, g["out"])
    print(f"DiT from an fp16 safetensors file through HipOps vs the reference golden: rel-err {e:.3e}")
    assert e < 8e-3
    p8 = str(tmp_path / "seedvr2_ema_3b_fp8_e4m3fn.safetensors")
    sd8 = {k: (v.to(torch.float8_e4m3fn) if v.dim() >= 2 else v.to(torch.float16)).contiguous() for k, v in sd.items()}
    save_file({("model.diffusion_model." + k): v for k, v in sd8.items()}, p8)
    eng8, _ = ck.build_engines(hip, dit_path=p8, dit_cfg=cfg)
    out8 = eng8.forward(g["vid"].cuda(), txt.cuda(), 1000.0).float().cpu()
    up = {k: (v.to(BF16) if not k.endswith("freqs") else v) for k, v in sd8.items()}   # what autocast makes of the e4m3 weights
    want8 = dit_oracle.dit_forward(up, cfg, g["vid"], txt, 1000.0, windows_mod=windows)
    e8 = rel_err(out8, want8)
    print(f"DiT from an fp8-e4m3 safetensors file (ComfyUI prefix) through HipOps vs the oracle on the up-cast weights: {e8:.3e}; "
          f"fp8 weights move the output by {rel_err(want8, g['out']):.2e}")
    assert e8 < 8e-3
    assert rel_err(want8, g["out"]) > 2 * e8          # (the check can tell fp8 weights from bf16 ones)


def test_vae_checkpoint_files_through_hipops(hip, tmp_path):
    """fp16 VAE file -> encode / decode reproduce the reference goldens of vae_small.pt; a file that stores some kernels as 2D
    image-VAE weights is inflated ('tail', causal_inflation_lib.py:440-457) and must match the CPU oracle on the inflated dict."""
    from safetensors.torch import save_file
    from oracle import vae_oracle
    ck, weights, config = sub("checkpoint"), sub("weights"), sub("config")
    g = _golden("vae_small.pt")
    vcfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(vcfg, seed=g["seed_weights"])
    p16 = str(tmp_path / "ema_vae_fp16.safetensors")
    save_file({k: v.to(torch.float16).contiguous() for k, v in sd.items()}, p16)
    _, eng = ck.build_engines(hip, vae_path=p16)
    enc = eng.encode(g["x"][0].cuda()).float().cpu()               # scaled latent [T', h, w, 16] (infer.py:188)
    z = (g["z_in"][0].permute(1, 2, 3, 0).float() * vcfg.scaling_factor).to(BF16).cuda()
    dec = eng.decode(z).float().cpu()
    want_enc = g["enc"][0].permute(1, 2, 3, 0) * vcfg.scaling_factor
    pe, pd = rel_err(enc, want_enc), _psnr(dec, g["dec"][0], 2.0)
    print(f"VAE from an fp16 safetensors file through HipOps vs the reference goldens: encode rel-err {pe:.3e}, decode {pd:.1f} dB")
    assert pe < 1.2e-2 and pd >= 50.0
    flat = {k: (v[:, :, -1].contiguous() if v.dim() == 5 and v.shape[2] == 3 and ".conv1" in k else v.contiguous())
            for k, v in sd.items()}                                 # every resnet's conv1 stored as a 2D kernel
    p2d = str(tmp_path / "ema_vae_2d_fp16.safetensors")
    save_file(flat, p2d)
    _, eng2 = ck.build_engines(hip, vae_path=p2d)
    inflated = ck.prepare_vae_state_dict(ck.load_state_dict(p2d), vcfg)
    x = g["x"][0][:, :5, :32, :48].contiguous()
    got = eng2.encode(x.cuda()).float().cpu()
    want = vae_oracle.runner_vae_encode(x.float(), inflated, vcfg)
    e2 = rel_err(got, want)
    print(f"VAE with 2D-stored conv1 kernels (tail inflation) through HipOps vs the oracle: encode rel-err {e2:.3e}")
    assert e2 < 1.5e-2


# ------------------------------------------------------------------------------------------------------------------------
needs_reference = pytest.mark.skipif(not rl.available(), reason="needs the reference (checkout, or oracle/_ref built by oracle/build_ref.py)")


def _hip_runner(hip, g):
    config, weights, dit, vae, runner = (sub(n) for n in ("config", "weights", "dit", "vae", "runner"))
    dcfg, vcfg = config.DIT_TINY, config.VAEConfig(block_out_channels=tuple(g["vae_channels"]))
    r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg))
    r.dit = dit.NaDiTEngine(dcfg, weights.synth_dit_state_dict(dcfg, seed=g["seed_dit"]), hip)
    r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=g["seed_vae"]), hip)
    return r


def _run_reference_phases(ns, runner, images, text, g, seed=42):
    debug = rl.PhaseDebug()
    ctx = rl.phase_context("cuda:0", BF16, text.to(device="cuda:0", dtype=BF16))
    kw = dict(batch_size=g["batch_size"], temporal_overlap=g["temporal_overlap"])
    ctx = ns["encode_all_batches"](runner, ctx, images, debug, uniform_batch_size=g["uniform_batch_size"], seed=seed,
                                   resolution=g["resolution"], color_correction="lab", **kw)
    ctx = ns["upscale_all_batches"](runner, ctx, debug, seed=seed)
    ctx = ns["decode_all_batches"](runner, ctx, debug)
    ctx = ns["postprocess_all_batches"](ctx, debug, color_correction="lab", **kw)
    return ctx["final_video"]


@needs_reference
def test_reference_phase_functions_drive_the_hip_runner(hip):
    """The reference's phase functions over HIP engines == this repo's pipeline.upscale over the same engines (same seeds: both
    draw their noise from torch.cuda's generator in the same order), up to the bf16 glue between the runner calls."""
    weights, pipeline = sub("weights"), sub("pipeline")
    g = _golden("pipeline_small.pt")
    images = torch.rand(g["frames"], g["hw"][0], g["hw"][1], 3, generator=torch.Generator().manual_seed(g["seed_images"]))
    text = weights.synth_text_embedding()
    runner = _hip_runner(hip, g)
    got = _run_reference_phases(rl.reference_phases(), runner, images, text, g).float().cpu()
    want = pipeline.upscale(images.cuda(), runner, text.cuda(), resolution=g["resolution"], batch_size=g["batch_size"],
                            uniform_batch_size=g["uniform_batch_size"], temporal_overlap=g["temporal_overlap"],
                            color_correction="lab", seed=42, skip_trimmed_frames=False, output_dtype=None).float().cpu()
    assert got.shape == want.shape == tuple(g["out"].shape)
    d = (got - want).abs()
    e, q999 = rel_err(got, want), float(d.flatten().kthvalue(int(d.numel() * 0.999)).values)
    print(f"reference phases ({rl.kind()}) over the HIP runner vs pipeline.upscale over the same runner: rel-err {e:.2e}, "
          f"99.9 % of pixels within {q999:.2e}")
    assert e < 4e-3 and q999 < 8e-3                   # bf16 storage: single roundings of the glue (CPU double: 2.2e-3 / 3.9e-3)


@needs_reference
def test_reference_phase_functions_over_the_hip_runner_reproduce_the_reference_chain_golden(hip):
    """... and with the golden's noise injected where the phase code calls torch.randn_like, the reference's phases over the HIP
    engines reproduce the golden of the reference's own models (tests/golden/pipeline_small.pt) at the north star's bar."""
    from oracle import make_golden as mg
    weights = sub("weights")
    g = _golden("pipeline_small.pt")
    images = torch.rand(g["frames"], g["hw"][0], g["hw"][1], 3, generator=torch.Generator().manual_seed(g["seed_images"]))

    class TorchWithGoldenNoise:                      # what the phase functions see as `torch`
        def __init__(self):
            self.pending = []

        def __getattr__(self, name):
            return getattr(torch, name)

        def randn_like(self, t, **kw):
            if not self.pending:                     # upscale_all_batches draws base_noise, then the augmentation noise
                self.pending = list(mg.pipeline_noise(t))
            return self.pending.pop(0).to(device=t.device, dtype=kw.get("dtype", t.dtype))

    runner = _hip_runner(hip, g)
    out = _run_reference_phases(rl.reference_phases(TorchWithGoldenNoise()), runner, images, weights.synth_text_embedding(), g)
    out = out.float().cpu()
    p = _psnr(out, g["out"], 1.0)
    print(f"reference phases ({rl.kind()}) over the HIP runner vs the reference-chain golden: PSNR {p:.1f} dB at the nominal peak")
    assert p >= 50.0
