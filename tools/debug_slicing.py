"""GPU debug aid: where do temporally sliced and unsliced VAE encodes diverge?  (not part of the product)"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    config, weights, vae, ops = sub("config"), sub("weights"), sub("vae"), sub("ops")
    hip = ops.HipOps("cuda:0")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg), hip)
    g = torch.Generator().manual_seed(3)
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    x = (torch.rand(3, frames, size, size, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    if len(sys.argv) > 3:
        eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg), hip, trunk_store=sys.argv[3], branch_store=sys.argv[4])
    print(f"{frames} frames {size}x{size}, trunk {eng.trunk_store} branch {eng.branch_store}")

    rec = {}
    orig_conv, orig_gn = eng._conv, eng._gn

    def conv_hook(cw, xin, st, first, *a, **kw):
        out = orig_conv(cw, xin, st, first, *a, **kw)
        o, stats = out if isinstance(out, tuple) else (out, None)
        rec.setdefault(cw.name, []).append(ops.h16_to_float(o).clone())
        if stats is not None:
            rec.setdefault(cw.name + "  [fused stats]", []).append(stats.float().clone())
        return out

    gn_count = [0]

    def gn_hook(nm, xin, silu, stats=None):
        out = orig_gn(nm, xin, silu, stats)
        rec.setdefault(f"gn{gn_count[0]}", []).append(out.float().clone())
        gn_count[0] += 1
        return out

    eng._conv, eng._gn = conv_hook, gn_hook

    def run(fps):
        rec.clear()
        outs = []
        # GN hook names must line up across slices: reset the counter per slice via wrapper
        orig_slice = eng._encoder_slice

        def slice_hook(xs, st, first):
            gn_count[0] = 0
            return orig_slice(xs, st, first)

        eng._encoder_slice = slice_hook
        out = eng.encode(x, frames_per_slice=fps).float()
        eng._encoder_slice = orig_slice
        return out, {k: torch.cat(v, dim=0) for k, v in rec.items()}

    a, ra = run(1 << 20)
    a2, ra2 = run(1 << 20)
    b, rb = run(4)
    print(f"run-to-run (unsliced twice): {rel(a2, a):.3e}")
    print(f"sliced vs unsliced:          {rel(b, a):.3e}")
    for k in ra:
        if ra[k].shape != rb[k].shape:
            print(f"{k:60s} SHAPE {tuple(ra[k].shape)} vs {tuple(rb[k].shape)}")
            continue
        per_t = [rel(rb[k][t], ra[k][t]) for t in range(ra[k].shape[0])]
        print(f"{k:60s} rr {rel(ra2[k], ra[k]):.2e}  sliced {rel(rb[k], ra[k]):.2e}  per-frame "
              + " ".join(f"{v:.1e}" for v in per_t))


if __name__ == "__main__":
    main()
