"""Measurement builds only (SVR_BUILD_ABLATIONS=1): per-workgroup s_memtime stamps of the shipped conv kernel
(start / after prologue / after K loop / end) for one launch."""
import ctypes as C, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import sub
ops_mod, packing, hip_lib = sub("ops"), sub("packing"), sub("hip_lib")
ops = ops_mod.HipOps("cuda")
if os.environ.get("CONV_LDS"):
    ops.set_option("conv_lds", int(os.environ["CONV_LDS"]))
ROWS = int(os.environ.get("CONV_ROWS", "8"))
ops.set_option("conv_rows", ROWS)
T, H, W, Ci, Co = (int(v) for v in os.environ.get("SHAPE", "5,1024,1024,128,128").split(","))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(T, H, W, Ci, device="cuda", generator=g).bfloat16()
w = packing.pack_conv3d(torch.randn(Co, Ci, 3, 3, 3, generator=g, device="cuda") / math.sqrt(27 * Ci), "cuda")
b = torch.zeros(Co, device="cuda")
geom = ops_mod.Conv3dGeom(T, H, W, Ci, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
wf = ops.pack_conv_frag(w, 3, Ci, Co)
y = torch.empty(T, H, W, Co, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(x, w, y, N=Co, K=27 * Ci, bias=b, conv=geom, ldc=Co, W_frag=wf)
torch.cuda.synchronize()
ops.set_option("pipe_abl", 256)
ops.gemm(x, w, y, N=Co, K=27 * Ci, bias=b, conv=geom, ldc=Co, W_frag=wf)
torch.cuda.synchronize()
buf = np.zeros((4096, 4), dtype=np.uint64)
lib = hip_lib.lib()
lib.svr_debug_conv_timeline.argtypes = [C.c_void_p, C.c_int64]
assert lib.svr_debug_conv_timeline(buf.ctypes.data, buf.nbytes) == 0
t = buf.astype(np.int64)
d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]], 1)
first = d[:512]            # the first resident wave of workgroups (cold start), then steady state
late = d[2048:4096]
print(f"shape T,H,W,Cin,Cout = {T},{H},{W},{Ci},{Co}; {ROWS} rows per wave; {27 * Ci // 32} tap intervals per tile")
for name, v in (("first 512 workgroups", first), ("workgroups 2048..4095", late)):
    print(name, "median ticks: prologue %d, K loop %d, epilogue %d, total %d" % tuple(np.median(v, 0)))
    print("   p90: prologue %d, K loop %d, epilogue %d" % tuple(np.percentile(v[:, :3], 90, 0)))
ep = np.zeros((4096, 16), dtype=np.uint64)
lib.svr_debug_conv_epilogue.argtypes = [C.c_void_p, C.c_int64]
assert lib.svr_debug_conv_epilogue(ep.ctypes.data, ep.nbytes) == 0
e = ep.astype(np.int64)[2048:4096]
base = t[2048:4096, 2]
names = ["bias landed"] + [f"pass{p} {what}" for p in range(ROWS // 2) for what in ("LDS writes issued", "barrier passed", "stores issued")]
prev = base
for i, nme in enumerate(names):
    print("   epilogue +%-26s median %6d ticks" % (nme, np.median(e[:, i] - prev)))
    prev = e[:, i]
print("(ticks = shader clock cycles)")
