#!/usr/bin/env python
"""BASELINE config 1 IN FULL on the host's cores ("cfg-1 in full (DiT 32 layers + VAE enc/dec)"): the
REFERENCE's own NaDiT-3B (32 layers, PyTorch-SDPA path, fp32) and VideoAutoencoderKLWrapper -- imported by oracle/reference_loader.py
from the checkout or, on the GPU box, from oracle/_ref -- on a single 256 x 256 image (latent 1 x 32 x 32 -> 256 video + 58 text
tokens), 1 warm-up + median of 3 per leg, core count printed.  bench.py's in-run `cpu_baseline` times a bounded SAMPLE of the same
classes (5 x 96 x 96 clip + a 2-layer DiT: ~20 s) and converts by FLOPs, because building 3.4e9 random fp32 parameters twice (13.6 GB
each) takes longer than the whole GPU measurement; this script is the un-sampled figure next to it, run once per round on the GPU
box's host and committed under profiles/.       python tools/cpu_cfg1_full.py > profiles/rN_cpu_cfg1_full.json
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
    from oracle import reference_loader as rl
    assert rl.available(), "needs the reference checkout or oracle/_ref"
    config, weights, flops = sub("config"), sub("weights"), sub("flops")
    # (torch's default thread count = what bench.py's cpu_baseline uses; on a box whose cgroup grants fewer cores than os.cpu_count()
    # reports, forcing cpu_count() threads oversubscribes and the run never finishes)
    def note(msg):
        print(f"[cpu_cfg1_full +{time.perf_counter() - t00:6.1f}s] {msg}", file=sys.stderr, flush=True)
    t00 = time.perf_counter()
    import bench
    torch.set_num_threads(min(torch.get_num_threads(), bench.usable_cores()))   # (the cgroup quota, not the 256 logical CPUs the box reports)
    note(f"threads {torch.get_num_threads()}, usable cores {bench.usable_cores()}, os.cpu_count {os.cpu_count()}")
    g = torch.Generator().manual_seed(42)
    t_build = time.perf_counter()
    dcfg, vcfg = config.DIT_3B, config.VAE_V3
    dsd = {k: v.float() for k, v in weights.synth_dit_state_dict(dcfg).items()}
    note("3B weights drawn")
    ref_dit = rl.build_reference_dit(dcfg.as_dict(), dsd)
    del dsd
    note("reference NaDiT built")
    vsd = {k: v.float() for k, v in weights.synth_vae_state_dict(vcfg).items()}
    ref_vae = rl.build_reference_vae(vsd)
    t_build = time.perf_counter() - t_build
    x = torch.rand(3, 1, 256, 256, generator=g) * 2 - 1
    vid = torch.randn(1, 32, 32, 33, generator=g)
    txt = weights.synth_text_embedding().float()
    lat = {}

    def enc():
        with torch.no_grad():
            lat["z"] = ref_vae.encode(x[None]).latent

    def dec():
        with torch.no_grad():
            ref_vae.decode(lat["z"])

    def dit():
        with torch.no_grad():
            ref_dit(vid=vid.reshape(-1, 33), txt=txt, vid_shape=torch.tensor([[1, 32, 32]]), txt_shape=torch.tensor([[txt.shape[0]]]),
                    timestep=torch.tensor([1000.0]))

    note("models built")
    t_enc = median3(enc); note(f"encode {t_enc:.2f}s")
    t_dit = median3(dit); note(f"dit {t_dit:.2f}s")
    t_dec = median3(dec); note(f"decode {t_dec:.2f}s")
    fv = flops.vae_flops_tiled(vcfg, 1, 256, 256, False)
    fd = flops.dit_flops(dcfg, (1, 16, 16))["total"]
    total = t_enc + t_dit + t_dec
    f_cfg3 = flops.dit_flops(dcfg, (9, 135, 240))["total"] + sum(flops.vae_flops_tiled(vcfg, 33, 2160, 3840, True).values())
    res = {"what": "BASELINE config 1 in full on the host CPU: the reference's own NaDiT-3B (32 layers, SDPA path, fp32) + video VAE v3 "
                   f"({rl.kind()}: {rl.REFERENCE_ROOT if rl.kind() == 'source' else 'oracle/_ref'}), one 256 x 256 image",
           "cores": torch.get_num_threads(), "os_cpu_count": os.cpu_count(), "timing": "1 warm-up + median of 3 per leg",
           "seconds": {"vae_encode": t_enc, "dit_32_layers": t_dit, "vae_decode": t_dec, "total": total, "model_build": t_build},
           "algorithmic_tflop": {"vae_encode": fv["encode"] / 1e12, "dit": fd / 1e12, "vae_decode": fv["decode"] / 1e12},
           "cpu_tflops": (fv["encode"] + fv["decode"] + fd) / total / 1e12,
           "frames_per_s_cfg1": 1.0 / total,
           "frames_per_s_cfg3_extrapolated_by_flops": 32.0 / (f_cfg3 / ((fv["encode"] + fv["decode"] + fd) / total)),
           "note": "cfg-3 figure = 32 frames / (13.03 PFLOP algorithmic / this run's sustained FLOP rate): an extrapolation, labelled as such"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
