#!/usr/bin/env python
"""Gate 1 of the Winograd question: what do the stride-1 3x3 convs of the VAE cost in dB when they are
computed in a Winograd domain with 2-byte MFMA operands?  CPU only, zero GPU seconds.

The engines' host code runs over the torch double of the C ABI in the PRODUCT's storage regime (tests/ops_reference.py, bf16 MFMA
operands, h16 trunk / stream); every stride-1 conv with 3x3 spatial taps is replaced by one of

  spatial  F(2x2, 3x3): V = B^T d B in fp32 from the bf16 GroupNorm output, rounded to the operand type; U = G g G^T in fp32 from the
                        stored weights, rounded to the operand type; 16 channel contractions accumulated in fp32 (the MFMA), summed
                        over the temporal taps in the Winograd domain; Y = A^T M A in fp32.  2.25x fewer multiply-adds.
  temporal F(2, 3)    : pairs of output frames from four combinations of input frames (d0-d2, d1+d2, d2-d1, d1-d3) and four
                        combinations of the temporal taps (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2), each product a plain 1x3x3 conv;
                        1.5x fewer multiply-adds on the kt = 3 convs.  Frame 0 of a clip keeps the engine's two-term causal head.
  direct              : today's arithmetic with the operands re-rounded to the operand type (the control: "bf16" must reproduce
                        the shipped row; "fp16" prices 11-bit conv operands by themselves)

with operand type bf16 (8 significant bits), fp16 (11 bits) or fp32 (the transform's own rounding only).
Pass (the review's bar): >= 50.5 dB on pipeline_prod with one of the 2-byte operand types.

    python tools/winograd_budget.py [--fixture pipeline_prod] [--rows spatial,temporal,direct]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conftest import GOLDEN                       # noqa: E402
from ops_reference import TorchOps                # noqa: E402
import error_budget as eb                         # noqa: E402

BF16 = torch.bfloat16
BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])


def rnd(t, kind):
    if kind == "fp32":
        return t
    if kind == "fp16":
        return t.to(torch.float16).float()
    return t.to(BF16).float()


def wino2d(x, w, kind):
    """x [B, C, H, W] (already zero-padded by 1), w [N, C, 3, 3] -> [B, N, H-2, W-2] by F(2x2, 3x3), operands rounded to ``kind``."""
    B, C, Hp, Wp = x.shape
    Ho, Wo = Hp - 2, Wp - 2
    x = F.pad(x, (0, Wo % 2, 0, Ho % 2))
    d = x.unfold(2, 4, 2).unfold(3, 4, 2)                                    # [B, C, ny, nx, 4, 4]
    V = rnd(torch.einsum("ij,bcyxjk,lk->bcyxil", BT, d, BT), kind)
    U = rnd(torch.einsum("ij,ncjk,lk->ncil", G, w, G), kind)
    M = torch.einsum("ncil,bcyxil->bnyxil", U, V)
    Y = torch.einsum("ij,bnyxjk,lk->bnyxil", AT, M, AT)                      # [B, N, ny, nx, 2, 2]
    ny, nx = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, 2 * ny, 2 * nx)[:, :, :Ho, :Wo]


class WinoOps(TorchOps):
    def __init__(self, mode, kind):
        super().__init__("cpu", act_dtype=BF16)
        self.mode, self.kind, self.hits, self.macs_direct, self.macs_done = mode, kind, 0, 0.0, 0.0

    def gemm(self, A, W, out, *, N, K, conv=None, phase=None, gn_groups=0, **kw):
        g = conv
        if (g is None or phase is not None or tuple(g.k[1:]) != (3, 3) or tuple(g.stride) != (1, 1, 1) or g.Cin < 16
                or tuple(g.pad[1:]) != (1, 1)):
            return super().gemm(A, W, out, N=N, K=K, conv=conv, phase=phase, gn_groups=gn_groups, **kw)
        kt = g.k[0]
        x = A.reshape(g.T, g.H, g.W, g.Cin).float()
        pt = g.pad[0]
        if pt > 0:
            head = g.halo.float()[-pt:] if g.halo is not None else x[:1].expand(pt, g.H, g.W, g.Cin)
            x = torch.cat([head, x], dim=0)
        assert x.shape[0] == g.To + kt - 1, (x.shape, g)
        xp = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1))                      # [T', C, H+2, W+2]
        w = W[:N, :K].float().reshape(N, kt, 3, 3, g.Cin).permute(1, 0, 4, 2, 3)   # [kt, N, C, 3, 3]
        self.hits += 1
        vox = g.To * g.H * g.W * N * g.Cin
        self.macs_direct += vox * kt * 9
        if self.mode == "spatial":
            y = sum(wino2d(xp[i:i + g.To], w[i], self.kind) for i in range(kt))
            self.macs_done += vox * kt * 4
        elif self.mode == "temporal" and kt == 3 and g.To >= 2:
            k = self.kind
            c2 = lambda xx, ww: F.conv2d(rnd(xx, k), rnd(ww, k))
            ys = []
            for t in range(0, g.To - 1, 2):
                d0, d1, d2, d3 = xp[t], xp[t + 1], xp[t + 2], xp[t + 3]
                m0 = c2((d0 - d2)[None], w[0])
                m1 = c2((d1 + d2)[None], (w[0] + w[1] + w[2]) * 0.5)
                m2 = c2((d2 - d1)[None], (w[0] - w[1] + w[2]) * 0.5)
                m3 = c2((d1 - d3)[None], w[2])
                ys += [m0 + m1 + m2, m1 - m2 - m3]
            self.macs_done += (g.To // 2 * 2) * g.H * g.W * N * g.Cin * 2 * 9
            if g.To % 2:
                t = g.To - 1
                ys.append(sum(F.conv2d(rnd(xp[t + i][None], k), rnd(w[i], k)) for i in range(3)))
                self.macs_done += g.H * g.W * N * g.Cin * 27
            y = torch.cat(ys, 0)
        else:
            k = self.kind
            y = sum(F.conv2d(rnd(xp[i:i + g.To], k), rnd(w[i], k)) for i in range(kt))
            self.macs_done += vox * kt * 9
        acc = y.permute(0, 2, 3, 1).reshape(-1, N)                           # [To*H*W, N]
        # the epilogue (bias, residual, store) is the double's own: hand the accumulator over as a 1x1 "GEMM" with identity weights
        return self._epilogue(acc, out, N=N, gn_groups=gn_groups, **kw)

    def _epilogue(self, acc, out, *, N, gn_groups, bias=None, epilogue=0, gate=None, resid=None, out_f32=False, **kw):
        from ops_reference import _ld, _st, EPI_RESID_GATE
        res = acc
        if bias is not None:
            res = res + bias[:N].float()
        if epilogue == EPI_RESID_GATE:
            if gate is not None:
                res = res * gate[:N].float()
            if resid is not None:
                res = res + _ld(resid.reshape(res.shape[0], -1)[:, :N])
        else:
            assert epilogue == 0, epilogue
        out.reshape(res.shape[0], -1)[:, :N].copy_(_st(res, out))
        return (out, None) if gn_groups > 0 else out


def run_vae17(g, mg, ops):
    """decode of tests/golden/vae_tiled17.pt (17 frames 96 x 160, tiled 64 / 32: every decoder conv, seconds of CPU)."""
    config, weights, vae = eb.sub("config"), eb.sub("weights"), eb.sub("vae")
    cfg = config.VAE_V3
    eng = vae.VideoVAEEngine(cfg, weights.synth_vae_state_dict(cfg, seed=g["seed_weights"]), ops)
    kw = dict(tiled=True, tile_size=tuple(g["tile_size"]), tile_overlap=tuple(g["tile_overlap"]))
    z = (mg.latent_input(*g["latent"], seed=g["seed_z"])[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(BF16)
    y = eng.decode(z, **kw).float()
    return eb.rel_err(y, g["dec_tiled"][0]), eb.psnr_nominal(y, g["dec_tiled"][0], 2.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="pipeline_prod", help="pipeline_prod | pipeline_small | vae_tiled17 (decode only, fast)")
    ap.add_argument("--rows", default="direct,spatial,temporal")
    ap.add_argument("--kinds", default="bf16,fp16,fp32")
    args = ap.parse_args()
    from oracle import make_golden as mg
    g = torch.load(os.path.join(GOLDEN, args.fixture + ".pt"), weights_only=True)
    if args.fixture.startswith("pipeline"):
        run = lambda ops: eb.run_pipeline(None, g, mg, vae_ops=ops)
    else:
        run = lambda ops: run_vae17(g, mg, ops if ops is not None else TorchOps("cpu", act_dtype=BF16))
    e, p = run(None)
    print(f"{args.fixture:14s} shipped regime (TorchOps, bf16 operands)                               rel-err {e:.3e}  PSNR(nominal) {p:6.2f} dB", flush=True)
    for mode in args.rows.split(","):
        for kind in args.kinds.split(","):
            ops = WinoOps(mode, kind)
            e, p = run(ops)
            print(f"{args.fixture:14s} {mode:8s} operands {kind:5s} ({ops.hits:4d} convs, multiply-adds x{ops.macs_done / max(ops.macs_direct, 1):.3f})"
                  f"   rel-err {e:.3e}  PSNR(nominal) {p:6.2f} dB", flush=True)


if __name__ == "__main__":
    main()
