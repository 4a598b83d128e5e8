#!/usr/bin/env python
"""Where do the dB go?  Per-site error budget of the bf16 storage regime against the reference's fp32 goldens, on the CPU.

The engines' host code runs over the torch double of the C ABI (tests/ops_reference.py: fp32 arithmetic on the stored values,
i.e. what the MFMA kernels compute up to summation order) with every activation held in fp32, and each *class of store* is
rounded to bf16 or left exact by a switch.  One line per experiment: all stores rounded (= the product today), then each
class left exact in turn (what fixing THAT store would buy), then cumulative candidates.  Test/measurement tooling only.

    python tools/error_budget.py [--fixture vae_tiled17|pipeline_small] [--quick]
    python tools/error_budget.py --fixture pipeline_prod --fp8      # BASELINE config 5's "fp8 MFMA weight path", priced (see Fp8DitOps)
"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sub, rel_err, GOLDEN            # noqa: E402
from ops_reference import TorchOps, EPI_RESID_GATE, EPI_SWIGLU   # noqa: E402

BF16 = torch.bfloat16

# store classes ------------------------------------------------------------------------------------------------------
VAE_SITES = ("trunk",        # conv / GEMM outputs that carry a residual (ResnetBlock3D output, attention output): the skip path
             "branch",       # conv outputs inside a block (conv1 -> norm2), conv_in, upsampler outputs, shortcut convs
             "gn_out",       # GroupNorm-apply(+SiLU) outputs = the bf16 A operand of the next conv (MFMA input: must round)
             "attn",         # mid-block attention internals (q, k, v, P, P V)
             "io")           # latent scaling in, blended tile out
DIT_SITES = ("hid",          # the residual stream (gate+residual epilogues)
             "norm_out",     # RMSNorm+modulation outputs = GEMM A operands (must round)
             "qkv", "att", "mlp_hid", "dit_io")


class BudgetOps(TorchOps):
    """TorchOps with fp32 allocations and a per-class rounding switch."""

    def __init__(self, rounded):
        super().__init__("cpu", act_dtype=torch.float32)
        self.rounded = set(rounded)

    def _r(self, t, site):
        if site in self.rounded and t.dtype == torch.float32:
            t.copy_(t.to(BF16).float())
        return t

    def gemm(self, A, W, out, *, N, K, epilogue=0, resid=None, conv=None, out_f32=False, gn_groups=0, **kw):
        r = super().gemm(A, W, out, N=N, K=K, epilogue=epilogue, resid=resid, conv=conv, out_f32=out_f32, gn_groups=gn_groups, **kw)
        o = r[0] if gn_groups > 0 else r
        if out_f32 and conv is None and kw.get("bias") is None and resid is None:
            return r                                    # attention scores: a real fp32 store (every tensor is fp32 here, so the
                                                        # engines' own out_f32 = (out.dtype == fp32) says nothing)
        if conv is not None:
            site = "trunk" if resid is not None else "branch"
        elif self._dit:
            site = ("hid" if epilogue == EPI_RESID_GATE else "mlp_hid" if epilogue in (EPI_SWIGLU, 4) else
                    "qkv" if N % 3 == 0 and N // 3 % 128 == 0 and resid is None and N > K else "dit_io")
        else:
            site = "trunk" if resid is not None else "attn"
        self._r(o, site)
        return r

    _dit = False

    def rmsnorm_mod(self, x, out, *a, **k):
        return self._r(super().rmsnorm_mod(x, out, *a, **k), "norm_out")

    def qknorm_rope(self, qkv, *a, **k):
        return self._r(super().qknorm_rope(qkv, *a, **k), "qkv")

    def attn_varlen(self, qkv, out, *a, **k):
        return self._r(super().attn_varlen(qkv, out, *a, **k), "att" if self._dit else "attn")

    def rows_mean(self, src, dst, *a, **k):
        return self._r(super().rows_mean(src, dst, *a, **k), "att")

    def softmax_rows(self, S, P, scale):
        return self._r(super().softmax_rows(S, P, scale), "attn")

    def patchify(self, vid, out):
        return self._r(super().patchify(vid, out), "dit_io")

    def unpatchify_euler(self, pred, x_t, out):
        return self._r(super().unpatchify_euler(pred, x_t, out), "dit_io")

    def groupnorm_apply(self, x, out, *a, **k):
        return self._r(super().groupnorm_apply(x, out, *a, **k), "gn_out")

    def blend_finalize(self, acc, cnt, out, *a, **k):
        return self._r(super().blend_finalize(acc, cnt, out, *a, **k), "io")

    def affine_slice(self, inp, out, *a, **k):
        return self._r(super().affine_slice(inp, out, *a, **k), "io")


E4M3 = torch.float8_e4m3fn


def quant_e4m3(x: torch.Tensor, mode: str) -> torch.Tensor:
    """fp32 values -> what an e4m3 MFMA operand would hold (returned as fp32), K along the last axis.
    "tensor": one scale amax / 448 per tensor;  "row": one per row (token / output channel);  "mx": the OCP MX format of
    v_mfma_scale_f32_16x16x128_f8f6f4 -- blocks of 32 along K share a power-of-two scale 2^(floor(log2 amax) - 8);
    "raw": no scale at all (how the reference's fp8 checkpoints hold the weights, compatibility.py:895-938)."""
    x = x.float()
    if mode == "raw":
        return x.clamp(-448, 448).to(E4M3).float()
    if mode == "mx":
        k = x.shape[-1]
        pad = (-k) % 32
        xb = torch.nn.functional.pad(x, (0, pad)).reshape(*x.shape[:-1], -1, 32)
        amax = xb.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -120)
        sc = torch.exp2(torch.floor(torch.log2(amax)) - 8)
        q = ((xb / sc).clamp(-448, 448).to(E4M3).float() * sc).reshape(*x.shape[:-1], -1)
        return q[..., :k]
    amax = (x.abs().amax() if mode == "tensor" else x.abs().amax(-1, keepdim=True)).clamp_min(1e-30)
    sc = amax / 448.0
    return (x / sc).to(E4M3).float() * sc


class Fp8DitOps(TorchOps):
    """The product's storage regime (bf16 stores; the NaDiT's residual stream as NaDiTEngine defaults it -- h16 since round 5, or
    what SVR_DIT_STREAM says: run_pipeline prints neither, so quote the default with the numbers) with the operands of the NaDiT's four big GEMMs per block (qkv,
    attn-out, mlp-in, mlp-out: every GEMM with K >= 2048 that sees the video tokens) rounded to e4m3 the way an fp8 MFMA path would
    have to: ``act`` / ``wgt`` in (None, "tensor", "row", "mx", "raw").  Arithmetic stays fp32 (the MFMA accumulates in fp32)."""

    def __init__(self, act, wgt):
        super().__init__("cpu", act_dtype=BF16)
        self.act, self.wgt, self.hits = act, wgt, 0

    def gemm(self, A, W, out, *, N, K, conv=None, **kw):
        if conv is None and K >= 2048 and N >= 2048 and A.shape[0] > 64:
            self.hits += 1
            if self.act:
                A = quant_e4m3(A.reshape(-1, A.shape[-1])[:, :K], self.act)
            if self.wgt:
                W = quant_e4m3(W[:N, :K], self.wgt)
        return super().gemm(A, W, out, N=N, K=K, conv=conv, **kw)


def psnr_nominal(a, b, peak):
    mse = float((a.double() - b.double()).pow(2).mean())
    return 10 * math.log10(peak * peak / max(mse, 1e-30))


def run_vae17(rounded, g, sd, mg, **eng_kw):
    config, vae = sub("config"), sub("vae")
    cfg = config.VAE_V3
    ops = BudgetOps(rounded) if rounded is not None else TorchOps("cpu", act_dtype=BF16)     # None: the product's storage regime
    eng = vae.VideoVAEEngine(cfg, sd, ops, **eng_kw)
    kw = dict(tiled=True, tile_size=tuple(g["tile_size"]), tile_overlap=tuple(g["tile_overlap"]))
    z = (mg.latent_input(*g["latent"], seed=g["seed_z"])[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(BF16).float()
    y = eng.decode(z, **kw).float()
    return rel_err(y, g["dec_tiled"][0]), psnr_nominal(y, g["dec_tiled"][0], 2.0)


def run_pipeline(rounded, g, mg, dit_ops=None, vae_ops=None, **eng_kw):
    config, weights, dit, vae, runner, pipeline = (sub(n) for n in ("config", "weights", "dit", "vae", "runner", "pipeline"))
    dcfg, vcfg = getattr(config, g.get("dit", "DIT_TINY")), config.VAEConfig(block_out_channels=tuple(g["vae_channels"]))
    tile = (dict(encode_tiled=True, decode_tiled=True, encode_tile_size=tuple(g["vae_tile"]), decode_tile_size=tuple(g["vae_tile"]),
                 encode_tile_overlap=tuple(g["vae_tile_overlap"]), decode_tile_overlap=tuple(g["vae_tile_overlap"]))
            if g.get("vae_tile") else {})
    if dit_ops is not None:
        ops_v, ops_d = TorchOps("cpu", act_dtype=BF16), dit_ops
    elif rounded is None:
        ops_v = ops_d = TorchOps("cpu", act_dtype=BF16)
    else:
        ops_v, ops_d = BudgetOps(rounded), BudgetOps(rounded)
        ops_d._dit = True
    if vae_ops is not None:
        ops_v = vae_ops
    r = runner.VideoDiffusionInfer(runner.default_config(dcfg, vcfg), **tile)
    dit_kw = {k: eng_kw.pop(k) for k in ("hid_fp32",) if k in eng_kw}
    r.dit = dit.NaDiTEngine(dcfg, weights.synth_dit_state_dict(dcfg, seed=g["seed_dit"]), ops_d, **dit_kw)
    sample_dtype = eng_kw.pop("sample_dtype", None)
    r.vae = vae.VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, seed=g["seed_vae"]), ops_v, **eng_kw)
    r.vae.sample_dtype = sample_dtype
    images = torch.rand(g["frames"], g["hw"][0], g["hw"][1], 3, generator=torch.Generator().manual_seed(g["seed_images"]))
    out = pipeline.upscale(images, r, weights.synth_text_embedding().float(), resolution=g["resolution"],
                           batch_size=g["batch_size"], uniform_batch_size=g["uniform_batch_size"],
                           temporal_overlap=g["temporal_overlap"], color_correction="lab", noise_provider=mg.pipeline_noise).float()
    return rel_err(out, g["out"]), psnr_nominal(out, g["out"], 1.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="vae_tiled17")
    ap.add_argument("--quick", action="store_true", help="only: everything rounded / candidates")
    ap.add_argument("--engine-only", action="store_true", help="only the rows that run the engines in their real storage regimes")
    ap.add_argument("--rows", default="", help="comma-separated substrings: only the rows whose name contains one of them")
    ap.add_argument("--fp8", action="store_true",
                    help="pipeline fixtures: the shipped regime with the operands of the NaDiT's big GEMMs rounded to e4m3 (Fp8DitOps)")
    args = ap.parse_args()
    from oracle import make_golden as mg
    weights, config = sub("weights"), sub("config")
    g = torch.load(os.path.join(GOLDEN, args.fixture + ".pt"), weights_only=True)
    if args.fixture.startswith("pipeline"):
        sites = VAE_SITES + DIT_SITES
        run = lambda rounded, **kw: run_pipeline(rounded, g, mg, **kw)
    else:
        sites = VAE_SITES
        sd = weights.synth_vae_state_dict(config.VAE_V3, seed=g["seed_weights"])
        run = lambda rounded, **kw: run_vae17(rounded, g, sd, mg, **kw)
    allr = set(sites)
    if args.fp8:
        for name, act, wgt in (("shipped regime (bf16 operands)", None, None),
                               ("weights e4m3 unscaled (a fp8 checkpoint as the reference up-casts it), bf16 activations", None, "raw"),
                               ("weights e4m3 per-row scale, bf16 activations", None, "row"),
                               ("activations e4m3 per-tensor scale, bf16 weights", "tensor", None),
                               ("activations e4m3 per-token scale, bf16 weights", "row", None),
                               ("activations e4m3 MX blocks of 32 (v_mfma_scale f8f6f4), bf16 weights", "mx", None),
                               ("both operands e4m3 MX blocks of 32", "mx", "mx"),
                               ("both operands e4m3 per-row / per-token scale", "row", "row")):
            ops_d = Fp8DitOps(act, wgt)
            e, p = run(None, dit_ops=ops_d)
            print(f"{args.fixture:16s} fp8: {name:95s} rel-err {e:.3e}   PSNR(nominal) {p:6.2f} dB   ({ops_d.hits} GEMMs)", flush=True)
        return
    rows = [(f"ENGINE store trunk={t} branch={b}" + (" (round 4 as shipped)" if (t, b) == ("h16", "h16") else ""), None, dict(trunk_store=t, branch_store=b))
            for t, b in (("fp32", "h16"), ("h16", "h16"), ("h16", "bf16"), ("fp32", "fp32"))]
    rows += [("ENGINE sample fp32, store trunk=h16 branch=h16", None, dict(trunk_store="h16", branch_store="h16", sample_dtype=torch.float32)),
             ("ENGINE sample fp32, store trunk=fp32 branch=bf16", None, dict(trunk_store="fp32", branch_store="bf16", sample_dtype=torch.float32)),
             ("ENGINE glue fp32 only, store trunk=fp32 branch=bf16", None, dict(trunk_store="fp32", branch_store="bf16"))]
    rows += [("ENGINE, bf16 storage, trunk_fp32=False (round 2)", None, dict(trunk_fp32=False)),
            ("ENGINE, bf16 storage, trunk_fp32=True, branch_fp32=False (round 3 as shipped)", None, dict(trunk_fp32=True, branch_fp32=False)),
            ("ENGINE, bf16 storage, trunk_fp32=True, branch_fp32=True (round 3 with its +2.1 % option)", None, dict(trunk_fp32=True, branch_fp32=True)),
            ("every store bf16 (product)", allr, {}), ("no store rounded (weights bf16 only, sub-pixel merge)", set(), {}),
            ("no store rounded, two-step upsamplers + three-tap head", set(), dict(merge_upsamplers=False, merge_causal_head=False))]
    if not args.quick:
        rows += [(f"exact: {s}", allr - {s}, {}) for s in sites]
        rows += [(f"ONLY rounded: {s}", {s}, {}) for s in sites]
    rows += [("exact: trunk + branch (only MFMA operands rounded)", allr - {"trunk", "branch", "hid"}, {}),
             ("exact: trunk + hid", allr - {"trunk", "hid"}, {}),
             ("every store bf16, two-step upsamplers", allr, dict(merge_upsamplers=False))]
    for name, rounded, kw in rows:
        if args.engine_only and rounded is not None:
            continue
        if args.rows and not any(t in name for t in args.rows.split(",")):
            continue
        e, p = run(rounded, **dict(kw))
        print(f"{args.fixture:16s} {name:70s} rel-err {e:.3e}   PSNR(nominal) {p:6.2f} dB", flush=True)


if __name__ == "__main__":
    main()
