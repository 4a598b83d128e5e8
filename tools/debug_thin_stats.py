"""GPU debug aid: fused GroupNorm statistics of the conv kernels per output kind -- are all partial slots written, are the
reduced statistics bit-reproducible?  (not part of the product)"""
import ctypes as C
import importlib
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


def main():
    ops, packing, hip_lib = sub("ops"), sub("packing"), sub("hip_lib")
    hip = ops.HipOps("cuda:0")
    lib = hip.lib
    g = torch.Generator(device="cuda").manual_seed(1)
    T, H, W = 5, 512, 512
    for name, Cin, Cout in (("thin conv_in", 4, 128), ("3x3x3 128->128", 128, 128)):
        x = (torch.randn(T, H, W, Cin, generator=g, device="cuda")).to(torch.bfloat16)
        if Cin == 4:
            x[..., 3] = 0
            Wp = packing.pack_conv3d((torch.randn(Cout, 3, 3, 3, 3, generator=g, device="cuda") / 9).to(torch.bfloat16), "cuda", 4)
            frag = None
        else:
            Wp = packing.pack_conv3d((torch.randn(Cout, Cin, 3, 3, 3, generator=g, device="cuda") / math.sqrt(27 * Cin)).to(torch.bfloat16), "cuda")
            frag = hip.pack_conv_frag(Wp, 3, Cin, Cout)
        geom = ops.Conv3dGeom(T, H, W, Cin, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
        bias = torch.randn(Cout, generator=g, device="cuda")
        for kind, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32), ("h16", torch.float16)):
            res = []
            for rep in range(3):
                out = torch.empty(T, H, W, Cout, device="cuda", dtype=dt)
                a, M = ops.fill_gemm_args(x, Wp, out, N=Cout, K=Wp.shape[1], bias=bias, conv=geom, ldc=Cout, W_frag=frag,
                                          out_f32=dt != torch.bfloat16, zeros_ptr=hip.zeros.data_ptr(), chk=hip._chk)
                a.gn_groups = 32
                nblk = int(lib.svr_gemm_gn_blocks(C.byref(a)))
                partial = torch.full((T * nblk * 32 * 2,), float("nan"), dtype=torch.float64, device="cuda")
                a.gn_partial = partial.data_ptr()
                cls = hip_lib.KERNEL_CLASSES[int(lib.svr_gemm_kernel_class(C.byref(a)))]
                hip_lib.check(lib.svr_gemm_bf16(C.byref(a), hip._stream()), "gemm")
                torch.cuda.synchronize()
                res.append((partial.clone(), out.clone()))
            nan = int(torch.isnan(res[0][0]).sum())
            same_p = all(torch.equal(res[0][0].view(torch.int64), r[0].view(torch.int64)) for r in res[1:])
            same_o = all(torch.equal(res[0][1], r[1]) for r in res[1:])
            d = (res[0][0] - res[1][0]).abs()
            print(f"{name:16s} [{cls}] out {kind}: nblk {nblk}, unwritten partial slots {nan} of {res[0][0].numel()}, partials reproducible "
                  f"{same_p} (max diff {float(d[~torch.isnan(d)].max()) if d.numel() else 0:.3e}), output reproducible {same_o}")


if __name__ == "__main__":
    main()
