#!/usr/bin/env python
"""A/B of the persistent GEMM kernels on the NaDiT's four shapes, every variant checked bit for bit against gemm_w4q_kernel first:
    gemm_w4r = 0  both operands through LDS (gemm_w4q_kernel, round 3)
    gemm_w4r = 1  gemm_w4r_kernel: weights straight into registers, activations into LDS by LDS-DMA (default)
  plus, with the measurement build (SVR_BUILD_ABLATIONS=1), svr_set_option("pipe_abl", v) variants given as --abl v,v,...
  python tools/w4r_variants.py [--reps 5] [--abl 701,702]"""
import argparse, importlib, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = "comfyui-seedvr2_videoupscaler_amd"
ops_mod, packing = importlib.import_module(pkg + ".ops"), importlib.import_module(pkg + ".packing")


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--abl", default="")
    ap.add_argument("--M", type=int, default=291600)
    args = ap.parse_args()
    ops = ops_mod.HipOps("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    M = args.M
    variants = [("w4q (gemm_w4r=0)", 0, 0), ("gemm_w4r=1", 1, 0)]
    variants += [(f"gemm_w4r=1 pipe_abl={v}", 1, int(v)) for v in filter(None, args.abl.split(","))]
    for name, N, K, epi, f32 in (("qkv 2560->7680", 7680, 2560, ops_mod.EPI_BIAS, False),
                                 ("attn-out 2560->2560 (+gate, fp32 stream in place)", 2560, 2560, ops_mod.EPI_RESID_GATE, True),
                                 ("mlp-in swiglu 2560->2x6912", 13824, 2560, ops_mod.EPI_SWIGLU, False),
                                 ("mlp-out 6912->2560 (+gate, fp32 stream in place)", 2560, 6912, ops_mod.EPI_RESID_GATE, True)):
        a = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = packing.pack_matrix(torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K), "cuda")
        wf = ops.pack_gemm_frag(w)
        nout = N // 2 if epi == ops_mod.EPI_SWIGLU else N
        hid0 = torch.rand(M, nout, generator=g, device="cuda", dtype=torch.float32) * 2 - 1 if f32 else None
        gate = torch.rand(N, generator=g, device="cuda", dtype=torch.float32)
        want = None
        for label, w4r, abl in variants:
            ops.set_option("gemm_w4r", w4r)
            ops.set_option("pipe_abl", abl)

            def run(out):
                if f32:
                    ops.gemm(a, w, out, N=N, K=K, epilogue=epi, gate=gate, resid=out, out_f32=True, W_frag=wf)
                else:
                    ops.gemm(a, w, out, N=N, K=K, epilogue=epi, W_frag=wf)
            out = hid0.clone() if f32 else torch.empty(M, nout, device="cuda", dtype=torch.bfloat16)
            run(out)
            torch.cuda.synchronize()
            if want is None:
                want = out.clone()
            same = bool(torch.equal(out, want))
            scratch = hid0.clone() if f32 else out
            sec = timeit(lambda: run(scratch), args.reps)
            print(json.dumps({"gemm": name, "variant": label, "us": round(sec * 1e6, 1), "tflops": round(2.0 * M * N * K / sec / 1e12, 1),
                              "bit_identical_to_w4q": same}), flush=True)
        ops.set_option("gemm_w4r", 1); ops.set_option("pipe_abl", 0)
        del a, w, wf, hid0, want


if __name__ == "__main__":
    main()
