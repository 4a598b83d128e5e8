#!/usr/bin/env python
"""Kernel micro-benchmark at the hot path's real shapes (cfg3): one line per kernel with HIP-event
timings and algorithmic TFLOP/s or GB/s.  Also the target of the rocprofv3 --pmc passes
(tools/gpu_pmc.sh), so kernel names / launch counts here are what profiles/ refers to.

    python tools/kbench.py [--reps 5] [--only conv,gemm,attn,side]
"""
import argparse
import importlib
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")
BF16 = torch.bfloat16


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="conv,gemm,attn,side")
    ap.add_argument("--match", default="", help="conv: only shapes whose name contains this substring")
    ap.add_argument("--attn-default-only", action="store_true", help="attn: skip the A/B build variants of the window kernel")
    ap.add_argument("--no-vae-attn", action="store_true", help="attn: only the DiT window attention")
    ap.add_argument("--no-frag", action="store_true", help="conv: do not supply the fragment-ordered weight copy")
    args = ap.parse_args()
    only = set(args.only.split(","))
    ops_mod, packing = sub("ops"), sub("packing")
    ops = ops_mod.HipOps("cuda:0")
    dev = ops.device
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, generator=g, device=dev) * 2 - 1).to(BF16)
    out = []

    def report(name, sec, flops=None, bytes_=None):
        r = {"kernel": name, "us": round(sec * 1e6, 1)}
        if flops:
            r["tflops"] = round(flops / sec / 1e12, 1)
        if bytes_:
            r["gbps"] = round(bytes_ / sec / 1e9, 1)
        out.append(r)
        print(json.dumps(r), flush=True)

    if "conv" in only:
        # (name, T, H, W, Cin, Cout): the shapes that carry the VAE FLOPs at cfg3 (1024-px tiles)
        for name, T, H, W, Ci, Co in (("conv3x3x3 128->128 @1024^2", 5, 1024, 1024, 128, 128),
                                      ("conv3x3x3 256->128 @1024^2", 5, 1024, 1024, 256, 128),
                                      ("conv3x3x3 256->256 @512^2", 9, 512, 512, 256, 256),
                                      ("conv3x3x3 512->256 @512^2", 9, 512, 512, 512, 256),
                                      ("conv3x3x3 512->512 @256^2", 9, 256, 256, 512, 512),
                                      # non power-of-two row pitch (HBM channel / TLB aliasing probe)
                                      ("conv3x3x3 128->128 @1024x1056", 5, 1024, 1056, 128, 128),
                                      ("conv3x3x3 128->128 @1024x992", 5, 1024, 992, 128, 128),
                                      ("conv3x3x3 256->256 @512x544", 9, 512, 544, 256, 256),
                                      ("conv3x3x3 128->3 @1024^2 (decoder conv_out)", 5, 1024, 1024, 128, 3)):
            if args.match and args.match not in name:
                continue
            x = rnd(T, H, W, Ci)
            w = packing.pack_conv3d(torch.randn(Co, Ci, 3, 3, 3, generator=g, device=dev) / math.sqrt(27 * Ci), dev)
            b = torch.zeros(Co, dtype=torch.float32, device=dev)
            y = ops.empty(T, H, W, Co)
            geom = ops_mod.Conv3dGeom(T, H, W, Ci, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
            wf = None if args.no_frag else ops.pack_conv_frag(w, 3, Ci, Co)
            sec = timeit(lambda: ops.gemm(x, w, y, N=Co, K=27 * Ci, bias=b, conv=geom, ldc=Co, W_frag=wf), args.reps)
            report(name, sec, flops=2.0 * T * H * W * Co * 27 * Ci)
            if Co <= 16:                                    # the 32-cout kernel these launches ran on until round 4
                ops.set_option("conv_thinout4", 0)
                sec = timeit(lambda: ops.gemm(x, w, y, N=Co, K=27 * Ci, bias=b, conv=geom, ldc=Co, W_frag=wf), args.reps)
                ops.set_option("conv_thinout4", 1)
                report(name + " [conv_thinout4=0: the 32-cout kernel]", sec, flops=2.0 * T * H * W * Co * 27 * Ci)
            del x, w, y
    if "thin" in only:
        # encoder.conv_in: RGB padded to 4 channels -> 128, the im2col image of a patch built in LDS; output on the h16 trunk with
        # fused GroupNorm statistics (how the engine launches it)
        T, H, W, Co = 9, 1024, 1024, 128
        x = rnd(T, H, W, 4)
        w = packing.pack_conv3d(torch.randn(Co, 3, 3, 3, 3, generator=g, device=dev) / 9.0, dev, 4)
        b = torch.zeros(Co, dtype=torch.float32, device=dev)
        geom = ops_mod.Conv3dGeom(T, H, W, 4, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
        for name, dt in (("h16", torch.float16), ("bf16", torch.bfloat16)):
            y = torch.empty(T, H, W, Co, device=dev, dtype=dt)
            sec = timeit(lambda: ops.gemm(x, w, y, N=Co, K=w.shape[1], bias=b, conv=geom, ldc=Co, gn_groups=32,
                                          out_f32=dt != torch.bfloat16), args.reps)
            report(f"thin conv_in 4->128 @9x1024^2, {name} output + fused statistics", sec, bytes_=T * H * W * (Co * 2 + 8))
            del y
    if "gemm" in only:
        M = 291600
        for name, N, K, epi in (("gemm qkv 2560->7680", 7680, 2560, ops_mod.EPI_BIAS),
                                ("gemm attn-out 2560->2560 (+gate,resid)", 2560, 2560, ops_mod.EPI_RESID_GATE),
                                ("gemm mlp-in swiglu 2560->2x6912", 13824, 2560, ops_mod.EPI_SWIGLU),
                                ("gemm mlp-out 6912->2560 (+gate,resid)", 2560, 6912, ops_mod.EPI_RESID_GATE)):
            a = rnd(M, K)
            w = packing.pack_matrix(torch.randn(N, K, generator=g, device=dev) / math.sqrt(K), dev)
            nout = N // 2 if epi == ops_mod.EPI_SWIGLU else N
            c = ops.empty(M, nout)
            kw = {}
            if epi == ops_mod.EPI_RESID_GATE:
                kw = dict(gate=torch.ones(N, dtype=torch.float32, device=dev), resid=rnd(M, N))
            sec = timeit(lambda: ops.gemm(a, w, c, N=N, K=K, epilogue=epi, **kw), args.reps)
            report(name, sec, flops=2.0 * M * N * K)
            wf = ops.pack_gemm_frag(w)        # gemm_w4r_kernel: weights from the fragment-ordered copy straight into registers, activations by LDS-DMA
            sec = timeit(lambda: ops.gemm(a, w, c, N=N, K=K, epilogue=epi, W_frag=wf, **kw), args.reps)
            report(name + " [W_frag: gemm_w4r_kernel]", sec, flops=2.0 * M * N * K)
            if epi == ops_mod.EPI_RESID_GATE:
                # the form the NaDiT engine issues (dit.py, wide residual stream): hid fp32, updated in place
                hid = torch.rand(M, N, generator=g, device=dev, dtype=torch.float32) * 2 - 1
                sec = timeit(lambda: ops.gemm(a, w, hid, N=N, K=K, epilogue=epi, gate=kw["gate"], resid=hid, out_f32=True), args.reps)
                report(name.replace("(+gate,resid)", "(+gate, fp32 stream in place)"), sec, flops=2.0 * M * N * K)
                sec = timeit(lambda: ops.gemm(a, w, hid, N=N, K=K, epilogue=epi, gate=kw["gate"], resid=hid, out_f32=True, W_frag=wf), args.reps)
                report(name.replace("(+gate,resid)", "(+gate, fp32 stream in place)") + " [W_frag: gemm_w4r_kernel]", sec, flops=2.0 * M * N * K)
                del hid
            del a, w, c, kw, wf
    if "ksweep" in only:
        # per-K-tile cost c and per-output-tile overhead o of the big-GEMM main loops: M x N = 4096 tiles of 256 x 256 (16 full rounds of
        # 256 CUs), bias epilogue, K swept; time per round = (K / 64) c + o.  The vendor library on the same shapes for calibration.
        import torch.nn.functional as F
        M, N = 65536, 4096
        for K in (512, 1024, 2048, 4096, 8192):
            a = rnd(M, K)
            wt = torch.randn(N, K, generator=g, device=dev) / math.sqrt(K)
            w = packing.pack_matrix(wt, dev)
            c = ops.empty(M, N)
            b = torch.zeros(N, dtype=torch.float32, device=dev)
            rounds = (M // 256) * (N // 256) / 256
            for opt, label in ((0, "gemm_kernel"), (1, "gemm_w4q")):
                ops.set_option("gemm_w4", opt)
                sec = timeit(lambda: ops.gemm(a, w, c, N=N, K=K, bias=b), args.reps)
                report(f"ksweep {label} K={K}", sec, flops=2.0 * M * N * K)
                out[-1]["us_per_round"] = round(sec * 1e6 / rounds, 2)
            ops.set_option("gemm_w4", 1)
            wb = wt.to(BF16)
            sec = timeit(lambda: F.linear(a, wb), args.reps)
            report(f"ksweep hipBLASLt K={K}", sec, flops=2.0 * M * N * K)
            out[-1]["us_per_round"] = round(sec * 1e6 / rounds, 2)
            del a, w, c, wb, wt
        fit = {}
        for r in out:
            if r["kernel"].startswith("ksweep"):
                _, label, k = r["kernel"].split()
                fit.setdefault(label, []).append((int(k[2:]) // 64, r["us"] / 16.0))
        for label, pts in fit.items():
            n = len(pts)
            sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
            sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
            cc = (n * sxy - sx * sy) / (n * sxx - sx * sx)
            print(json.dumps({"fit": label, "us_per_k_tile": round(cc, 4), "us_per_output_tile": round((sy - cc * sx) / n, 3),
                              "k_loop_tflops": round(2.0 * 256 * 256 * 64 * 256 / cc / 1e6, 1)}), flush=True)
    if "blaslt" in only:
        # calibration only (never on the product path): the vendor library's GEMM (hipBLASLt behind torch) on the same four shapes, same box
        import torch.nn.functional as F
        M = 291600
        for name, N, K in (("qkv 2560->7680", 7680, 2560), ("attn-out 2560->2560", 2560, 2560),
                           ("mlp-in 2560->13824", 13824, 2560), ("mlp-out 6912->2560", 2560, 6912)):
            a, w = rnd(M, K), rnd(N, K)
            sec = timeit(lambda: F.linear(a, w), args.reps)
            report("hipBLASLt (torch F.linear, no epilogue) " + name, sec, flops=2.0 * M * N * K)
            del a, w
    if "shortk" in only:
        # short-K problems of the VAE (tools/shape_census.py): pixel-shuffle upsamplers, 1x1 shortcut convs, attention scores
        for name, F_, H, W, Cc, rz in (("upsample gemm+pixel-shuffle 256->1024 @5x512^2", 5, 512, 512, 256, 1),
                                       ("upsample gemm+pixel-shuffle 512->4096 @3x256^2", 3, 256, 256, 512, 2)):
            x = rnd(F_ * H * W, Cc)
            w = packing.pack_matrix(torch.randn(4 * rz * Cc, Cc, generator=g, device=dev) / math.sqrt(Cc), dev)
            b = torch.zeros(4 * rz * Cc, dtype=torch.float32, device=dev)
            y = ops.empty(F_ * rz, 2 * H, 2 * W, Cc)
            ps = ops_mod.PixelShuffleGeom(F_, H, W, rz, Cc, False)
            sec = timeit(lambda: ops.gemm(x, w, y, N=4 * rz * Cc, K=Cc, M=F_ * H * W, bias=b, ps=ps), args.reps)
            report(name, sec, flops=2.0 * F_ * H * W * 4 * rz * Cc * Cc, bytes_=(x.numel() + y.numel()) * 2)
            del x, w, y
        for name, T, H, W, Ci, Co in (("conv1x1x1 256->128 @5x1024^2 (shortcut)", 5, 1024, 1024, 256, 128),
                                      ("conv1x1x1 512->256 @5x512^2 (shortcut)", 5, 512, 512, 512, 256)):
            x = rnd(T, H, W, Ci)
            w = packing.pack_conv3d(torch.randn(Co, Ci, 1, 1, 1, generator=g, device=dev) / math.sqrt(Ci), dev)
            b = torch.zeros(Co, dtype=torch.float32, device=dev)
            y = ops.empty(T, H, W, Co)
            geom = ops_mod.Conv3dGeom(T, H, W, Ci, T, H, W, (1, 1, 1), (1, 1, 1), (0, 0, 0), None)
            sec = timeit(lambda: ops.gemm(x, w, y, N=Co, K=Ci, bias=b, conv=geom, ldc=Co), args.reps)
            report(name, sec, flops=2.0 * T * H * W * Co * Ci, bytes_=(x.numel() + y.numel()) * 2)
            del x, w, y
        n = 16384
        q, k = rnd(n, 512), rnd(n, 512)
        S = torch.empty(n, n, dtype=torch.float32, device=dev)
        sec = timeit(lambda: ops.gemm(q, k, S, N=n, K=512, out_f32=True), args.reps)
        report("attention scores 16384x16384x512 (fp32 store)", sec, flops=2.0 * n * n * 512, bytes_=n * n * 4)
        del q, k, S
        for name, T, H, W, Ci, Co, k3, st in (("conv3x3x3 s2 256->256 @9x512^2 (downsample)", 9, 512, 512, 256, 256, (3, 3, 3), (2, 2, 2)),
                                              ("conv1x3x3 s(1,2,2) 128->128 @9x1024^2 (downsample)", 9, 1024, 1024, 128, 128, (1, 3, 3), (1, 2, 2))):
            x = rnd(T, H, W, Ci)
            w = packing.pack_conv3d(torch.randn(Co, Ci, *k3, generator=g, device=dev) / math.sqrt(k3[0] * 9 * Ci), dev)
            b = torch.zeros(Co, dtype=torch.float32, device=dev)
            pt = k3[0] - 1
            To, Ho, Wo = (T + pt - k3[0]) // st[0] + 1, (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
            y = ops.empty(To, Ho, Wo, Co)
            geom = ops_mod.Conv3dGeom(T, H, W, Ci, To, Ho, Wo, k3, st, (pt, 0, 0), None)
            sec = timeit(lambda: ops.gemm(x, w, y, N=Co, K=k3[0] * 9 * Ci, bias=b, conv=geom, ldc=Co), args.reps)
            report(name, sec, flops=2.0 * To * Ho * Wo * Co * k3[0] * 9 * Ci)
            del x, w, y
    if "subpixel" in only:
        # the sub-pixel upsampler convs: one phase launch each, low-resolution input -> 2x grid
        for name, T, H, W, C, kt, ts in (("sub-pixel conv (3,2,2) 256->256 @5x512^2 -> 1024^2 (spatial upsampler)", 5, 512, 512, 256, 3, 1),
                                         ("sub-pixel conv (2,2,2) 512->512 @5x256^2 -> 512^2 (temporal upsampler)", 5, 256, 256, 512, 2, 2),
                                         ("sub-pixel conv (2,2,2) 512->512 @5x128^2 -> 256^2 (temporal upsampler)", 5, 128, 128, 512, 2, 2)):
            x = rnd(T, H, W, C)
            halo = rnd(kt - 1, H, W, C)
            w = packing.pack_conv3d(torch.randn(C, C, kt, 2, 2, generator=g, device=dev) / math.sqrt(4 * kt * C), dev)
            wf = None if args.no_frag else ops.pack_conv_frag(w, kt, C, C, taps=(2, 2))
            b = torch.zeros(C, dtype=torch.float32, device=dev)
            bb = torch.zeros(3, C, dtype=torch.float32, device=dev)
            y = ops.empty(T * ts, 2 * H, 2 * W, C)
            geom = ops_mod.Conv3dGeom(T, H, W, C, T, H, W, (kt, 2, 2), (1, 1, 1), (kt - 1, 1, 1), halo)
            sec = timeit(lambda: ops.gemm(x, w, y, N=C, K=4 * kt * C, bias=b, conv=geom, W_frag=wf,
                                          phase=ops_mod.PhaseScatter(0, 0, bb, ts)), args.reps)
            report(name, sec, flops=2.0 * T * H * W * C * C * 4 * kt)
            if wf is not None:
                # the four spatial phases: as four launches (how rounds 2-3 ran them) and as ONE quad launch (phase fastest in the tile order)
                quad = [(py, px, w, b, bb, wf) for py in (0, 1) for px in (0, 1)]

                def four():
                    for py, px, *_ in quad:
                        gq = ops_mod.Conv3dGeom(T, H, W, C, T, H, W, (kt, 2, 2), (1, 1, 1), (kt - 1, 1 - py, 1 - px), halo)
                        ops.gemm(x, w, y, N=C, K=4 * kt * C, bias=b, conv=gq, W_frag=wf, phase=ops_mod.PhaseScatter(py, px, bb, ts))
                sec = timeit(four, args.reps)
                report(name + " -- all four phases, four launches", sec, flops=4 * 2.0 * T * H * W * C * C * 4 * kt)
                sec = timeit(lambda: ops.gemm(x, w, y, N=C, K=4 * kt * C, bias=b, conv=geom, W_frag=wf,
                                              phase=ops_mod.PhaseScatter(0, 0, bb, ts, quad=quad)), args.reps)
                report(name + " -- all four phases, ONE quad launch", sec, flops=4 * 2.0 * T * H * W * C * C * 4 * kt)
            del x, w, y
    if "attn" in only:
        windows, config = sub("windows"), sub("config")
        for method in ("720pwin_by_size_bysize", "720pswin_by_size_bysize"):
            size, Lt, heads = (9, 135, 240), 58, 20
            plan = windows.plan_windows(size, (4, 3, 3), method)
            N = size[0] * size[1] * size[2]
            import numpy as np
            lens = np.diff(plan.cu)
            seq = np.concatenate([np.concatenate([plan.tok[plan.cu[i]:plan.cu[i + 1]], N + np.arange(Lt)])
                                  for i in range(plan.n_win)]).astype(np.int32)
            outr = np.concatenate([np.concatenate([plan.tok[plan.cu[i]:plan.cu[i + 1]], N + Lt + i * Lt + np.arange(Lt)])
                                   for i in range(plan.n_win)]).astype(np.int32)
            cu = np.concatenate([[0], np.cumsum(lens + Lt)]).astype(np.int32)
            qkv = rnd(N + Lt, 3 * heads * 128)
            att = ops.empty(N + Lt + plan.n_win * Lt, heads * 128)
            t_seq, t_out, t_cu = (torch.from_numpy(v).to(dev) for v in (seq, outr, cu))
            fl = sum(4.0 * heads * 128 * float(l + Lt) ** 2 for l in lens)
            for impl, variant, tag in ((0, 0, "gen2 8 waves (default)"), (0, 1, "gen2 4 waves + setprio"), (0, 4, "gen2 4 waves"),
                                       (0, 3, "gen2 8 waves + setprio"), (1, 0, "gen1 svr_attn")):
                if variant and args.attn_default_only:
                    continue
                ops.set_option("attn_impl", impl)
                ops.set_option("attn_variant", variant)
                sec = timeit(lambda: ops.attn_varlen(qkv, att, t_seq, t_out, t_cu, int(lens.max()) + Lt, heads, 128,
                                                     1 / math.sqrt(128)), args.reps)
                report(f"attn window {method} ({plan.n_win} windows) [{tag}]", sec, flops=fl)
            ops.set_option("attn_impl", 0)
            ops.set_option("attn_variant", 0)
            del qkv, att
    if "attn" in only and not args.no_vae_attn:
        # VAE mid-block attention on one 1024-px tile: 9 frames x (128*128) tokens, 1 head of 512
        T, n, Cc = 9, 128 * 128, 512
        qkv = rnd(T * n, 3 * Cc)
        att = ops.empty(T * n, Cc)
        rows = torch.arange(T * n, dtype=torch.int32, device=dev)
        cu = (torch.arange(T + 1, dtype=torch.int32, device=dev) * n).contiguous()
        sec = timeit(lambda: ops.attn_varlen(qkv, att, rows, rows, cu, n, 1, Cc, 1 / math.sqrt(Cc)), max(1, args.reps // 2))
        report("attn VAE mid (9 x 16384 tokens, d=512)", sec, flops=4.0 * Cc * n * n * T)
        del qkv, att
        # the same attention as the VAE engine runs it at tile sizes: Q K^T (fp32 scores) -> row softmax -> P V, one frame
        q, k, v = rnd(n, Cc), rnd(n + 256, Cc), rnd(n, Cc)
        S = ops.empty(n, n, dtype=torch.float32)
        P = ops.empty(n, n)
        o = ops.empty(n, Cc)
        vt = v.t().contiguous()
        sec = timeit(lambda: ops.gemm(q, k[:n], S, N=n, K=Cc, out_f32=True), args.reps)
        report("VAE mid attn, 1 frame: S = Q K^T (16384^2 x 512, fp32 out)", sec, flops=2.0 * n * n * Cc)
        sec = timeit(lambda: ops.softmax_rows(S, P, 1 / math.sqrt(Cc)), args.reps)
        report("VAE mid attn, 1 frame: softmax rows 16384^2", sec, bytes_=n * n * 6)
        sec = timeit(lambda: ops.gemm(P, vt, o, N=Cc, K=n), args.reps)
        report("VAE mid attn, 1 frame: O = P V (16384 x 512 x 16384)", sec, flops=2.0 * n * n * Cc)
        sec = timeit(lambda: v.t().contiguous(), args.reps)
        report("VAE mid attn, 1 frame: V^T copy (torch)", sec, bytes_=n * Cc * 4)
        del q, k, v, S, P, o, vt
    if "side" in only:
        T, H, W, Cc = 5, 1024, 1024, 128
        x = rnd(T, H, W, Cc)
        y = ops.empty(T, H, W, Cc)
        stats = ops.empty(T, 32, 2, dtype=torch.float64)
        gam = torch.ones(Cc, dtype=torch.float32, device=dev)
        sec = timeit(lambda: ops.groupnorm_stats(x, stats, 32), args.reps)
        report("groupnorm_stats 5x1024^2x128", sec, bytes_=x.numel() * 2)
        sec = timeit(lambda: ops.groupnorm_apply(x, y, stats, gam, gam, 32, 1e-6, True), args.reps)
        report("groupnorm_apply+silu 5x1024^2x128", sec, bytes_=x.numel() * 4)
        for name, xw, nbytes in (("h16", (x.float() * 2.0 ** -6).to(torch.float16), 4), ("fp32", x.float(), 6)):     # wide trunk inputs
            sec = timeit(lambda: ops.groupnorm_apply(xw, y, stats, gam, gam, 32, 1e-6, True), args.reps)
            report(f"groupnorm_apply+silu 5x1024^2x128, {name} input", sec, bytes_=x.numel() * nbytes)
            sec = timeit(lambda: ops.groupnorm_stats(xw, stats, 32), args.reps)
            report(f"groupnorm_stats 5x1024^2x128, {name} input", sec, bytes_=x.numel() * (nbytes - 2))
        del x, y
        M, d = 291600, 2560
        x = rnd(M, d)
        y = ops.empty(M, d)
        sc = torch.ones(d, dtype=torch.float32, device=dev)
        sec = timeit(lambda: ops.rmsnorm_mod(x, y, 1e-5, scale=sc, shift=sc), args.reps)
        report("rmsnorm_mod 291600x2560", sec, bytes_=x.numel() * 4)
        xf = x.float()
        sec = timeit(lambda: ops.rmsnorm_mod(xf, y, 1e-5, scale=sc, shift=sc), args.reps)
        report("rmsnorm_mod 291600x2560, fp32 input (the NaDiT residual stream)", sec, bytes_=x.numel() * 6)
        del x, y, xf
        heads = 20
        qkv = rnd(M, 3 * heads * 128)
        pos = torch.randint(0, 60, (M, 3), device=dev, dtype=torch.int16)
        cos_t, sin_t = torch.rand(128, 21, device=dev), torch.rand(128, 21, device=dev)
        w = torch.ones(128, dtype=torch.float32, device=dev)
        sec = timeit(lambda: ops.qknorm_rope(qkv, heads, pos, 58, cos_t, sin_t, w, w, 1e-5), args.reps)
        report("qknorm_rope 291600 x 20 heads (q and k in place)", sec, bytes_=M * 2 * heads * 128 * 4)
    return out


if __name__ == "__main__":
    main()
