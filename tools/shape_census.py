"""Which GEMM / conv shapes does one VAE tile issue?  Runs the engine's host code over meta tensors with a recording
stand-in for the C-ABI ops (no arithmetic, no GPU) and prints the calls grouped by shape, with algorithmic FLOPs and
the minimum HBM bytes of each (inputs once + outputs once).  Used to decide which shapes deserve their own kernel."""
import collections
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


class MetaOps:
    act_dtype = torch.bfloat16
    device = torch.device("meta")

    def __init__(self, classify=False):
        """``classify``: ask the LIBRARY which kernel it would launch for every gemm call (svr_gemm_kernel_class on the
        svr_gemm_args the product's own ops.fill_gemm_args builds, with stand-in addresses): no GPU needed."""
        self.calls = collections.defaultdict(lambda: [0, 0.0, 0.0])
        self.classify = classify
        self.classes = collections.defaultdict(set)          # kind string -> kernel classes seen
        self.launches = []                                   # (kernel class, conv geometry | None, M, N, K)

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or torch.bfloat16, device="meta")

    def pack_conv_frag(self, W, kt, Cin, N, taps=(3, 3)):
        # (same acceptance rule as HipOps.pack_conv_frag)
        if Cin % 64 or N % 128 or W.shape[1] != kt * taps[0] * taps[1] * Cin or tuple(taps) not in ((3, 3), (2, 2)):
            return None
        return torch.empty(N * W.shape[1], dtype=torch.bfloat16, device="meta")

    def gemm(self, A, W, out, *, N, K, M=None, conv=None, ps=None, resid=None, out_f32=False, gn_groups=0, **kw):
        if self.classify:
            import ctypes
            ops_mod, hip_lib = sub("ops"), sub("hip_lib")
            kw2 = {k: v for k, v in kw.items() if k != "gn_shared"}
            a, _ = ops_mod.fill_gemm_args(A, W, out, N=N, K=K, M=M, conv=conv, ps=ps, resid=resid, out_f32=out_f32,
                                          zeros_ptr=0x1000, ptr=lambda t: 0x100000, **kw2)
            cls = hip_lib.KERNEL_CLASSES.get(int(hip_lib.lib().svr_gemm_kernel_class(ctypes.byref(a))), "invalid")
            self.launches.append((cls, conv, a.M, N, K))
        if conv is not None:
            g = conv
            M = g.To * g.Ho * g.Wo
            kind = f"conv k{g.k} s{g.stride} Cin{g.Cin}"
            in_bytes = g.T * g.H * g.W * g.Cin * 2
        else:
            M = M if M is not None else A.shape[0] if A.dim() == 2 else A.numel() // K
            kind = "gemm" + (" +pixel-shuffle" if ps is not None else "")
            in_bytes = M * K * 2
        out_bytes = M * N * (4 if out_f32 else 2) + (M * N * 2 if resid is not None else 0)
        key = (kind, M, N, K, "resid" if resid is not None else "", "f32" if out_f32 else "")
        c = self.calls[key]
        c[0] += 1
        c[1] += 2.0 * M * N * K
        c[2] += in_bytes + out_bytes + N * K * 2
        return (out, torch.empty(1, device="meta")) if gn_groups else out

    def __getattr__(self, name):                      # every other op: no arithmetic
        def f(*a, **k):
            self.calls[(name, 0, 0, 0, "", "")][0] += 1
            return None
        return f


def main():
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg, device="meta") if "device" in weights.synth_vae_state_dict.__code__.co_varnames \
        else weights.synth_vae_state_dict(cfg)
    ops = MetaOps()
    eng = vae.VideoVAEEngine(cfg, sd, ops)
    which = sys.argv[1] if len(sys.argv) > 1 else "decode"
    if which == "decode":
        eng.decode_clip(torch.empty(9, 128, 128, cfg.latent_channels, dtype=torch.bfloat16, device="meta"))
    else:
        eng.encode_clip(torch.empty(33, 1024, 1024, 4, dtype=torch.bfloat16, device="meta"))
    rows = sorted(ops.calls.items(), key=lambda kv: -kv[1][1])
    print(f"# {which} of one 1024x1024 tile (33 frames); kind | M | N | K | calls | GFLOP | min HBM MB | flop/byte")
    for (kind, M, N, K, r, f), (n, fl, by) in rows:
        if M == 0:
            continue
        print(f"{kind:34s} {M:10d} {N:6d} {K:6d} {r:5s} {f:3s} {n:4d} {fl / 1e9:10.1f} {by / 1e6:10.1f} {fl / max(by, 1):8.1f}")


if __name__ == "__main__":
    main()
