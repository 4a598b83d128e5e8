"""Debug helper: run the LDS-halo conv at a list of shapes with/without the fragment-ordered weights, sync + flush."""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import sub
ops_mod, packing = sub("ops"), sub("packing")
ops = ops_mod.HipOps("cuda")
if os.environ.get("CONV_ABL"):
    ops.set_option("pipe_abl", int(os.environ["CONV_ABL"]))
if os.environ.get("CONV_LDS"):
    ops.set_option("conv_lds", int(os.environ["CONV_LDS"]))
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(2, 128, 256, 128, 128), (5, 512, 512, 128, 128), (9, 256, 256, 512, 512)]
for T, H, W, Ci, Co in shapes:
    x = torch.randn(T, H, W, Ci, device="cuda", generator=g).bfloat16()
    w = packing.pack_conv3d(torch.randn(Co, Ci, 3, 3, 3, generator=g, device="cuda") / math.sqrt(27 * Ci), "cuda")
    b = torch.zeros(Co, device="cuda")
    geom = ops_mod.Conv3dGeom(T, H, W, Ci, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
    wf = ops.pack_conv_frag(w, 3, Ci, Co)
    torch.cuda.synchronize()
    print("packed", T, H, W, Ci, Co, flush=True)
    y0 = torch.empty(T, H, W, Co, device="cuda", dtype=torch.bfloat16)
    y1 = torch.empty_like(y0)
    ops.gemm(x, w, y0, N=Co, K=27 * Ci, bias=b, conv=geom, ldc=Co)
    torch.cuda.synchronize()
    print("  lds-weights ok", flush=True)
    for i in range(3):
        ops.gemm(x, w, y1, N=Co, K=27 * Ci, bias=b, conv=geom, ldc=Co, W_frag=wf)
        torch.cuda.synchronize()
        print("  wreg ok", i, bool(torch.equal(y0, y1)), flush=True)
