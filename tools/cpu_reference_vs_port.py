#!/usr/bin/env python
"""Calibrate bench.py's cpu_baseline ("kind": "port") against the REFERENCE ITSELF: the reference's model code (imported
unmodified from /root/reference through oracle/reference_loader.py) and the in-repo oracle port run the same sample on
the same host cores, 1 warm-up + median of 3 each.  Only runs where the reference is mounted
(the build container); writes profiles/r2_cpu_reference_vs_port.json.

    python tools/cpu_reference_vs_port.py > profiles/r2_cpu_reference_vs_port.json
"""
import importlib
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


def median3(fn):
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def main():
    from oracle import dit_oracle, vae_oracle, reference_loader as rl
    assert rl.available(), "needs /root/reference"
    config, weights, windows, flops = sub("config"), sub("weights"), sub("windows"), sub("flops")
    torch.set_num_threads(os.cpu_count())
    g = torch.Generator().manual_seed(0)
    # ---- the same sample bench.py's cpu_baseline uses
    vcfg = config.VAE_V3
    vsd = {k: v.float() for k, v in weights.synth_vae_state_dict(vcfg).items()}
    x = torch.rand(3, 5, 96, 96, generator=g) * 2 - 1
    ref_vae = rl.build_reference_vae(vsd)

    def port_vae():
        lat = vae_oracle.runner_vae_encode(x, vsd, vcfg)
        vae_oracle.runner_vae_decode(lat, vsd, vcfg)

    def ref_vae_run():
        with torch.no_grad():
            lat = ref_vae.encode(x[None]).latent
            ref_vae.decode(lat).sample

    dcfg = config.DiTConfig(num_layers=2, mm_layers=1)
    dsd = weights.synth_dit_state_dict(dcfg)
    vid = torch.randn(3, 48, 48, 33, generator=g)
    txt = weights.synth_text_embedding().float()
    ref_dit = rl.build_reference_dit(dcfg.as_dict(), {k: v.float() for k, v in dsd.items()})

    def ref_dit_run():
        with torch.no_grad():
            ref_dit(vid=vid.reshape(-1, 33), txt=txt, vid_shape=torch.tensor([[3, 48, 48]]), txt_shape=torch.tensor([[58]]),
                    timestep=torch.tensor([1000.0]))

    res = {"host_cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "timing": "1 warm-up + median of 3",
           "sample": "VAE enc+dec of a 5x96x96 clip (fp32) + 2-layer 3B-width DiT on a 3x48x48 latent (fp32)",
           "vae_seconds": {"reference": median3(ref_vae_run), "port": median3(port_vae)},
           "dit_seconds": {"reference": median3(ref_dit_run),
                           "port": median3(lambda: dit_oracle.dit_forward(dsd, dcfg, vid, txt, 1000.0, windows_mod=windows))}}
    f = sum(flops.vae_flops_tiled(vcfg, 5, 96, 96, False).values()) + flops.dit_flops(dcfg, (3, 24, 24))["total"]
    for kind in ("reference", "port"):
        t = res["vae_seconds"][kind] + res["dit_seconds"][kind]
        res[f"{kind}_tflops"] = f / t / 1e12
    res["port_over_reference_time"] = (res["vae_seconds"]["port"] + res["dit_seconds"]["port"]) / \
                                      (res["vae_seconds"]["reference"] + res["dit_seconds"]["reference"])
    res["note"] = ("bench.py's cpu_baseline times the PORT on the GPU box's host cores (the reference cannot travel there); this "
                   "file is the measured ratio between the port and the reference implementation on identical work, same machine")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
