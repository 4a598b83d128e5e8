"""GPU A/B of VideoVAEEngine(branch_fp32=): parity of the whole pipeline (pipeline_small) and of the tiled 17-frame decode (vae_tiled17)
against the reference's fp32 goldens with conv1 outputs stored in fp32 vs bf16.  usage: python tools/branch_ab.py"""
import sys, os, math, torch
sys.path.insert(0, "tests")
from conftest import sub, rel_err, GOLDEN
from oracle import make_golden as mg
config, weights, dit, vae, runner, pipeline, ops_mod = (sub(n) for n in ("config", "weights", "dit", "vae", "runner", "pipeline", "ops"))
hip = ops_mod.HipOps("cuda:0")
def pn(a, b, peak):
    return 10 * math.log10(peak * peak / float((a.double() - b.double()).pow(2).mean()))
g = torch.load(os.path.join(GOLDEN, "pipeline_small.pt"), weights_only=True)
g17 = torch.load(os.path.join(GOLDEN, "vae_tiled17.pt"), weights_only=True)
for branch in (True, False):
    dcfg, vcfg = config.DIT_TINY, config.VAEConfig(block_out_channels=tuple(g["vae_channels"]))
    r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg))
    r.dit = dit.NaDiTEngine(dcfg, weights.synth_dit_state_dict(dcfg, seed=g["seed_dit"]), hip)
    r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=g["seed_vae"]), hip, branch_fp32=branch)
    images = torch.rand(g["frames"], g["hw"][0], g["hw"][1], 3, generator=torch.Generator().manual_seed(g["seed_images"]))
    out = pipeline.upscale(images.cuda(), r, weights.synth_text_embedding().cuda(), resolution=g["resolution"], batch_size=g["batch_size"],
                           uniform_batch_size=g["uniform_batch_size"], temporal_overlap=g["temporal_overlap"], color_correction="lab",
                           noise_provider=mg.pipeline_noise).float().cpu()
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, seed=g17["seed_weights"]), hip, branch_fp32=branch)
    kw = dict(tiled=True, tile_size=tuple(g17["tile_size"]), tile_overlap=tuple(g17["tile_overlap"]))
    z = (mg.latent_input(*g17["latent"], seed=g17["seed_z"])[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(torch.bfloat16).cuda()
    y = eng.decode(z, **kw).float().cpu()
    print(f"branch_fp32={branch}: pipeline {pn(out, g['out'], 1.0):.2f} dB (rel-err {rel_err(out, g['out']):.3e}); vae_tiled17 decode {pn(y, g17['dec_tiled'][0], 2.0):.2f} dB")
