"""GPU timing of the phase glue around the three boundary calls at BASELINE config 4's shapes (17-frame batches, 720p -> 4K):
input transform, LAB colour correction, output conversion.  torch ops only (rows N1-N3 of SURVEY 8(f))."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")
pipeline, colorfix, transforms = sub("pipeline"), sub("colorfix"), sub("transforms")


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    frames = torch.rand(17, 720, 1280, 3, generator=g, device=dev)
    plans, _ = pipeline.plan_batches(17, 17, 0, False)
    out = {}
    out["prepare_batch (resize 720p->4K, pad, normalise), 17 frames"] = timeit(lambda: pipeline.prepare_batch(frames, plans[0], 2160, 0, dtype=torch.bfloat16))
    x = pipeline.prepare_batch(frames, plans[0], 2160, 0, dtype=torch.bfloat16).permute(1, 0, 2, 3).contiguous()   # T C H W
    y = (x.float() * 0.9 + 0.02 * torch.randn(x.shape, generator=g, device=dev)).to(torch.bfloat16)
    for m in ("lab", "wavelet", "adain"):
        out[f"colour correction '{m}', 17 frames 4K"] = timeit(lambda m=m: colorfix.METHODS[m](y, x), reps=2)
    out["clamp + [0,1] + THWC, 17 frames 4K"] = timeit(lambda: y.permute(0, 2, 3, 1).clamp(-1, 1).mul(0.5).add(0.5).to(torch.bfloat16))
    for k, v in out.items():
        print(json.dumps({"op": k, "ms": round(v * 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
