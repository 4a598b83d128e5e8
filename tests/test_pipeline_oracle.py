"""The whole four-phase chain against the golden produced by the REFERENCE's components (tests/golden/pipeline_small.pt:
the reference's NaDiT / VAE classes + the reference's text of the glue functions, driven by oracle/pipeline_oracle.py in
fp32; oracle/make_golden.py --only r2-pipe).  Two independent routes must land on it:
  * the product pipeline (pipeline.py + engines' host logic) over the fp32 torch double of the C ABI;
  * oracle/pipeline_oracle.py over the in-repo oracle models and glue restatements (what the GPU box can run).
The -m gpu counterpart is tests/test_gpu_parity.py::test_pipeline_vs_reference_golden."""
import os

import torch

from conftest import sub, rel_err, GOLDEN
from ops_reference import TorchOps


def _case():
    from oracle import make_golden as mg
    g = torch.load(os.path.join(GOLDEN, "pipeline_small.pt"), weights_only=True)
    images = torch.rand(g["frames"], g["hw"][0], g["hw"][1], 3, generator=torch.Generator().manual_seed(g["seed_images"]))
    return g, images, mg.pipeline_noise


def test_product_pipeline_fp32_double_equals_reference_chain():
    config, weights, dit, vae, runner, pipeline = (sub(n) for n in ("config", "weights", "dit", "vae", "runner", "pipeline"))
    g, images, noise = _case()
    dcfg, vcfg = config.DIT_TINY, config.VAEConfig(block_out_channels=tuple(g["vae_channels"]))
    ops = TorchOps("cpu", act_dtype=torch.float32)
    r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg))
    r.dit = dit.NaDiTEngine(dcfg, weights.synth_dit_state_dict(dcfg, seed=g["seed_dit"]), ops)
    # (the reference's two-step upsamplers: this test pins the GLUE exactly; the sub-pixel form rounds merged weights to bf16 and
    # is pinned by tests/test_host_logic.py and, against this same golden on the device, by tests/test_gpu_parity.py)
    r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=g["seed_vae"]), ops, merge_upsamplers=False, merge_causal_head=False)
    out = pipeline.upscale(images, r, weights.synth_text_embedding().float(), resolution=g["resolution"],
                           batch_size=g["batch_size"], uniform_batch_size=g["uniform_batch_size"],
                           temporal_overlap=g["temporal_overlap"], color_correction="lab", noise_provider=noise)
    assert out.shape == g["out"].shape
    d = (out.float() - g["out"]).abs()
    # the LAB transfer matches histograms by rank: an fp32 rounding difference can swap two ranks, so a handful of pixels
    # may move by a few per cent -- bound the bulk tightly and the outliers loosely
    assert rel_err(out.float(), g["out"]) < 5e-4 and float(d.flatten().kthvalue(int(d.numel() * 0.999)).values) < 2e-3
    assert float(d.max()) < 6e-2
    # skip_trimmed_frames: the padding frames the pipeline trims after decode are not decoded at all -- same clip (the golden's
    # batches carry uniform-batch and 4n+1 padding), fewer frames through the decoder
    decoded = []
    real = r.vae.decode
    r.vae.decode = lambda *a, **k: decoded.append(k.get("keep_frames")) or real(*a, **k)
    out2 = pipeline.upscale(images, r, weights.synth_text_embedding().float(), resolution=g["resolution"],
                            batch_size=g["batch_size"], uniform_batch_size=g["uniform_batch_size"],
                            temporal_overlap=g["temporal_overlap"], color_correction="lab", noise_provider=noise,
                            skip_trimmed_frames=True)
    r.vae.decode = real
    assert any(k is not None for k in decoded)
    d2 = (out2.float() - out.float()).abs()
    assert float(d2.flatten().kthvalue(int(d2.numel() * 0.999)).values) < 1e-4 and float(d2.max()) < 6e-2


def test_oracle_pipeline_over_in_repo_oracles_equals_reference_chain():
    from oracle import dit_oracle, vae_oracle, pipeline_oracle as po
    config, weights, windows, transforms, colorfix = (sub(n) for n in ("config", "weights", "windows", "transforms", "colorfix"))
    g, images, noise = _case()
    dcfg, vcfg = config.DIT_TINY, config.VAEConfig(block_out_channels=tuple(g["vae_channels"]))
    dsd = weights.synth_dit_state_dict(dcfg, seed=g["seed_dit"])
    vsd = weights.synth_vae_state_dict(vcfg, seed=g["seed_vae"])
    res = g["resolution"]
    comps = po.Components(
        pad_video_temporal=transforms.pad_video_temporal,
        video_transform=lambda x: transforms.video_transform(x, res, 0),
        true_target_dims=lambda h, w: transforms.true_target_dims(h, w, res, 0),
        vae_encode=lambda x: vae_oracle.runner_vae_encode(x.float(), vsd, vcfg),
        dit=lambda vid, text: dit_oracle.dit_forward(dsd, dcfg, vid, text, 1000.0, windows_mod=windows),
        vae_decode=lambda lat: vae_oracle.runner_vae_decode(lat.float(), vsd, vcfg),
        blend_overlapping_frames=transforms.blend_overlapping_frames,
        color_fix=lambda s, r: colorfix.lab_color_transfer(s, r, luminance_weight=0.8), noise=noise)
    out = po.upscale(images, weights.synth_text_embedding().float(), comps, g["batch_size"], g["temporal_overlap"],
                     g["uniform_batch_size"])
    assert out.shape == g["out"].shape and rel_err(out, g["out"]) < 5e-4
