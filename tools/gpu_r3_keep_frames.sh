#!/bin/bash
# Next round, first GPU call: VideoVAEEngine.decode(keep_frames=) / pipeline.upscale(skip_trimmed_frames=True) have only run on the CPU
# double of the C ABI -- check them on the HIP path (bit-equal to the leading frames of the full decode), then flip the pipeline default.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python - <<'PY' 2>&1 | tail -12
import sys, torch
sys.path.insert(0, "tests")
from conftest import sub
config, weights, vae_mod, ops_mod = sub("config"), sub("weights"), sub("vae"), sub("ops")
hip = ops_mod.HipOps("cuda:0")
cfg = config.VAE_V3
eng = vae_mod.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, device="cuda"), hip)
z = (torch.randn(4, 12, 10, cfg.latent_channels, device="cuda") * 0.5).to(torch.bfloat16)
for kw in ({}, dict(latents_per_slice=1), dict(tiled=True, tile_size=(64, 64), tile_overlap=(16, 16))):
    full = eng.decode(z, **kw)
    for k in (1, 2, 5, 9, 12, 13):
        y = eng.decode(z, keep_frames=k, **kw)
        y = y.unsqueeze(1) if y.dim() == 3 else y
        assert y.shape[1] == k and torch.equal(y, full[:, :k]), (kw, k)
print("decode(keep_frames) == decode()[:, :n] bit for bit on the HIP path")
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -s -k "pipeline or sharded" 2>&1 | grep -E "PSNR|passed|failed" | tail -4
