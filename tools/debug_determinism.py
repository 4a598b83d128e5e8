"""GPU debug aid: which op of a VAE encode / decode is not bit-reproducible run to run?  Every conv / GroupNorm / attention output
is reduced to an exact fingerprint (int64 sum of its raw bits viewed as int32/int16) and two identical runs are compared op by op.
usage: python tools/debug_determinism.py [frames] [size] [trunk_store] [branch_store] [encode|decode]   (not part of the product)"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


def fp(t):
    v = t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32 if t.element_size() == 4 else torch.int64)
    return int(v.to(torch.int64).sum()), int((v.to(torch.int64) * (torch.arange(v.numel(), device=v.device).reshape(v.shape) % 1021 + 1)).sum())


def main():
    config, weights, vae, ops = sub("config"), sub("weights"), sub("vae"), sub("ops")
    hip = ops.HipOps("cuda:0")
    cfg = config.VAE_V3
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    kw = dict(trunk_store=sys.argv[3], branch_store=sys.argv[4]) if len(sys.argv) > 4 else {}
    what = sys.argv[5] if len(sys.argv) > 5 else "encode"
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, device="cuda"), hip, **kw)
    g = torch.Generator(device="cuda").manual_seed(3)
    if what == "encode":
        x = (torch.rand(3, frames, size, size, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        run_it = lambda: eng.encode(x)
    else:
        z = torch.randn((frames - 1) // 4 + 1, size // 8, size // 8, 16, generator=g, device="cuda").to(torch.bfloat16)
        run_it = lambda: eng.decode(z)
    print(f"{what}: {frames} frames {size}x{size}, trunk {eng.trunk_store} branch {eng.branch_store}")
    rec = []
    orig_conv, orig_gn, orig_attn = eng._conv, eng._gn, eng._attention

    def conv_hook(cw, xin, st, first, *a, **k):
        out = orig_conv(cw, xin, st, first, *a, **k)
        o, stats = out if isinstance(out, tuple) else (out, None)
        rec.append((cw.name + f" -> {str(o.dtype)[6:]}{' +resid ' + str(k['resid'].dtype)[6:] if k.get('resid') is not None else ''}", fp(o)))
        if stats is not None:
            rec.append((cw.name + " [fused stats]", fp(stats)))
        return out

    def gn_hook(nm, xin, silu, stats=None):
        out = orig_gn(nm, xin, silu, stats)
        rec.append((f"groupnorm apply (input {str(xin.dtype)[6:]}, stats {'given' if stats is not None else 'separate pass'})", fp(out)))
        return out

    def attn_hook(ab, xin):
        out = orig_attn(ab, xin)
        rec.append((f"mid-block attention -> {str(out.dtype)[6:]}", fp(out)))
        return out

    eng._conv, eng._gn, eng._attention = conv_hook, gn_hook, attn_hook
    runs = []
    for _ in range(3):
        rec.clear()
        out = run_it()
        torch.cuda.synchronize()
        runs.append((list(rec), fp(out)))
    base = runs[0]
    print("ops:", len(base[0]))
    for i, (name, _) in enumerate(base[0][:400]):
        print(f"  #{i} {name}")

    for r, (ops_, final) in enumerate(runs[1:], 1):
        bad = [i for i, (a, b) in enumerate(zip(base[0], ops_)) if a != b]
        print(f"run {r} vs run 0: final {'EQUAL' if final == base[1] else 'DIFFERS'}; {len(bad)} of {len(ops_)} ops differ"
              + (f"; first: #{bad[0]} {ops_[bad[0]][0]}" if bad else ""))
        for i in bad[:6]:
            print(f"    #{i} {ops_[i][0]}")


if __name__ == "__main__":
    main()
