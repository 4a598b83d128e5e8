#!/usr/bin/env python
"""bench.py -- headline benchmark of the SeedVR2 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg1]

A "step" is one pass of the hot path over one temporal batch of synthetic input already resident in
HBM:  VAE encode -> one-step DiT (NaDiT-3B, 32 layers) -> VAE decode, i.e. exactly the three runner
calls the reference pipeline makes per batch (src/core/infer.py:117,315,203).  Workloads
(BASELINE.json configs):
    cfg3  33 frames 2160x3840 (720p->4K clip of 32 frames, 4n+1 padded), VAE tiled 1024/128   [metric config]
    cfg2   9 frames 2048x2048 (8-frame 512^2->2K clip), VAE untiled
    cfg1   1 frame  256x256
    cfg4  128 frames 720x1280 -> 2160x3840 through the whole four-phase pipeline (input transform, 8 temporal batches
          of 17 with a 1-frame overlap blend, LAB colour fix), batches dealt to the ranks (dist.upscale_sharded):
          BASELINE config 4, STRONG scaling (the clip is fixed, N ranks share its 8 batches)
N > 1: one process per GPU.  Launched under torchrun (RANK / WORLD_SIZE set) it joins that group; launched bare with
--gpus N it re-executes itself under torch.distributed.run with N ranks on 127.0.0.1 (and fails loudly when fewer
GPUs are visible).  cfg3/cfg2/cfg1/cfg5: every rank upscales its own temporal batch (weak scaling, no data-path
collective) and the upscaled bf16 frames are all-gathered over RCCL/xGMI inside the step.
Rank 0 prints ONE JSON line.  `value` = frames all ranks produced / max-over-ranks wall time; `n_gpus` is the
all-reduced count of ranks that ran.
--cpu-double (TESTS ONLY, refused unless SVR_BENCH_ALLOW_CPU_DOUBLE=1; tests/test_bench_launcher.py): the same script end to end --
self-launch under torch.distributed.run, rendezvous, the step, gather, guards, predicted_s, the JSON line -- on CPU ranks over gloo
with the torch double of the C ABI (tests/ops_reference.py) and reduced-width models, so that launcher plumbing cannot be what
fails the first multi-GPU run.  Its line carries "test_mode": true and its numbers mean nothing.
"""
import argparse
import importlib
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

# dmabuf IPC is the only mode the host driver supports: without it RCCL's first cross-process handle exchange fails with
# hipIpcGetMemHandle "invalid argument".  Set before the HIP runtime initialises (first CUDA call), for launchers that drop it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"

WORKLOADS = {
    # name: (frames, H, W, vae_tiled, description)
    "cfg3": (33, 2160, 3840, True, "SeedVR2-3B 32-frame 720p->4K clip (padded to 33 = 4n+1 for encode + DiT, 32 decoded), VAE tiled 1024/128"),
    "cfg2": (9, 2048, 2048, False, "SeedVR2-3B 8-frame 512^2->2048^2 clip (padded to 9 = 4n+1, 8 decoded), VAE untiled"),
    "cfg1": (1, 256, 256, False, "SeedVR2-3B single 256x256 image"),
    # BASELINE config 5's model and clip on ONE GPU, bf16 weights (the reference does no fp8 arithmetic either:
    # fp8 checkpoints are up-cast per op); not the metric config, run with --workload cfg5
    "cfg5": (65, 2160, 3840, True, "SeedVR2-7B 64-frame 1080p->4K clip (padded to 65 = 4n+1, 64 decoded), VAE tiled 1024/128"),
    # BASELINE config 4: the whole pipeline over a 128-frame clip, temporal batches sharded over the ranks (strong scaling)
    "cfg4": (128, 2160, 3840, True, "SeedVR2-3B 128-frame 720p->4K clip, 8 temporal batches of 17 (overlap 1) sharded "
                                    "over the ranks, full pipeline (transform, encode, DiT, decode, blend, LAB colour fix)"),
}
CFG4 = dict(in_hw=(720, 1280), resolution=2160, batch_size=17, temporal_overlap=1, uniform_batch_size=True,
            color_correction="lab")
PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16 peak, MI355X_MICROARCH.md


def sub(name):
    return importlib.import_module(f"{PKG}.{name}")


class _HostTick:
    """torch.cuda.Event's two methods on the host clock (--cpu-double only)."""
    def __init__(self, enable_timing=True):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def make_double_ops():
    """--cpu-double (tests only): the fp32 torch restatement of every C-ABI op (tests/ops_reference.py) with the recording
    interface of ProfiledOps, so that the rest of this script runs unchanged on CPU ranks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ops_reference import TorchOps

    class DoubleOps(TorchOps):
        def __init__(self):
            super().__init__("cpu")
            self.recording, self.events = False, []

        def gemm(self, A, W, out, **kw):
            t0 = time.perf_counter()
            r = super().gemm(A, W, out, **kw)
            if self.recording:
                self.events.append(("gemm", 2.0 * out.numel() * kw["K"], time.perf_counter() - t0))
            return r

        def summary(self):
            n, f, sec = len(self.events), sum(e[1] for e in self.events), sum(e[2] for e in self.events)
            return {"gemm": {"launches": n, "flops": f, "seconds": sec, "avg_us": sec / max(n, 1) * 1e6,
                             "tflops": f / max(sec, 1e-12) / 1e12}}

    return DoubleOps()


def make_profiled_ops(device):
    """HipOps whose conv implicit-GEMM launches are bracketed by HIP events recorded on the launch
    stream (torch's current stream IS the stream the C ABI launches on)."""
    ops_mod = sub("ops")

    class ProfiledOps(ops_mod.HipOps):
        def __init__(self, device):
            super().__init__(device)
            self.record_kernel_class = True
            self.recording = False
            self.events = []        # (kind, flops, start, end)

        def gemm(self, A, W, out, **kw):
            if not self.recording:
                return super().gemm(A, W, out, **kw)
            conv = kw.get("conv")
            if conv is not None:
                flops = 2.0 * conv.To * conv.Ho * conv.Wo * kw["N"] * conv.k[0] * conv.k[1] * conv.k[2] * conv.Cin
                if getattr(kw.get("phase"), "quad", None) is not None:
                    flops *= 4.0                         # one launch = the four spatial phases of a sub-pixel upsampler
            else:
                flops = 2.0 * (kw.get("M") or A.shape[0]) * kw["N"] * kw["K"]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()                                   # (on the CURRENT stream = the stream the C ABI launches on)
            r = super().gemm(A, W, out, **kw)
            e.record()
            # which kernel served the launch is the LIBRARY's answer (svr_gemm_kernel_class, the routing function the launch
            # itself uses: csrc/svr_gemm.hip gemm_route) -- "conv_halo" is conv_halo2_kernel alone
            if conv is not None:
                shape = (f"conv k{tuple(conv.k)} s{tuple(conv.stride)} out {conv.To}x{conv.Ho}x{conv.Wo} {conv.Cin}->{kw['N']}"
                         + (" quad" if getattr(kw.get("phase"), "quad", None) is not None else ""))
            else:
                shape = f"gemm M{kw.get('M') or A.shape[0]} N{kw['N']} K{kw['K']}" + (" +resid" if kw.get("resid") is not None else "")
            self.events.append((self.last_kernel_class, flops, s, e, shape))
            return r

        def by_shape(self, steps):
            """Per (kernel class, shape): launches per step, average duration, TFLOP/s, seconds per step -- the table the by-shape
            rocprof grouping cannot give (it has no FLOPs)."""
            agg = {}
            for kind, flops, s, e, shape in self.events:
                a = agg.setdefault((kind, shape), [0, 0.0, 0.0])
                a[0] += 1
                a[1] += flops
                a[2] += s.elapsed_time(e) * 1e-3
            rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
            tot = sum(v[2] for _, v in rows)
            lines = [f"# per-shape table of the GEMM / conv launches of one step (HIP events on the launch stream, {steps} timed steps; "
                     f"{sum(v[0] for _, v in rows) // steps} launches, {tot / steps * 1e3:.1f} ms per step)",
                     f"# {'kernel class':16s} {'shape':62s} {'n/step':>7s} {'avg us':>9s} {'TFLOP/s':>8s} {'ms/step':>8s} {'share':>6s}"]
            for (kind, shape), (n, f, sec) in rows:
                lines.append(f"{kind:18s} {shape:62s} {n // steps:7d} {sec / n * 1e6:9.1f} {f / max(sec, 1e-12) / 1e12:8.1f} "
                             f"{sec / steps * 1e3:8.2f} {100 * sec / max(tot, 1e-12):5.1f}%")
            return "\n".join(lines) + "\n"

        def summary(self):
            agg = {}
            for kind, flops, s, e, _shape in self.events:
                a = agg.setdefault(kind, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += flops
                a[2] += s.elapsed_time(e) * 1e-3
            return {k: {"launches": v[0], "flops": v[1], "seconds": v[2],
                        "avg_us": v[2] / v[0] * 1e6, "tflops": v[1] / max(v[2], 1e-12) / 1e12} for k, v in agg.items()}

    return ProfiledOps(device)


def usable_cores() -> int:
    """Cores this process may actually use: the scheduler affinity mask capped by the cgroup CPU quota (a GPU box reports 256 logical
    CPUs and hands the container a quota of 16: 128 torch threads on 16 cores run the reference ~3x SLOWER than 16 threads do)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline_cfg1() -> dict:
    """BASELINE config 1 IN FULL on this host's cores, in this run: the reference's own 32-layer NaDiT-3B (PyTorch-SDPA path, fp32) and
    VideoAutoencoderKLWrapper -- from the checkout or oracle/_ref -- on the IDENTICAL workload the GPU leg of `--workload cfg1` times
    (VAE encode of one 256 x 256 frame -> one-step DiT over the 1 x 32 x 32 latent + 58 text tokens -> VAE decode): 1 warm-up + median of
    3 of the whole chain.  Not a sample and not an extrapolation: both legs of the line ran the same shapes.  Test infrastructure used
    only as a reported baseline (generation_phases.py:171,542,807 make the same three runner calls per batch)."""
    from oracle import reference_loader as rl
    if not rl.available():
        return {"value": None, "unit": "frames/s", "kind": "unavailable", "sample": "neither the reference checkout nor oracle/_ref is present"}
    config, weights, flops = sub("config"), sub("weights"), sub("flops")
    torch.set_num_threads(min(torch.get_num_threads(), usable_cores()))
    cores = torch.get_num_threads()
    dcfg, vcfg = config.DIT_3B, config.VAE_V3
    t_build = time.perf_counter()
    dsd = weights.synth_dit_state_dict(dcfg)
    for k in list(dsd):                                  # (in place: 3.4e9 parameters are 13.6 GB in fp32)
        dsd[k] = dsd[k].float()
    ref_dit = rl.build_reference_dit(dcfg.as_dict(), dsd)
    del dsd
    ref_vae = rl.build_reference_vae({k: v.float() for k, v in weights.synth_vae_state_dict(vcfg).items()})
    t_build = time.perf_counter() - t_build
    g = torch.Generator().manual_seed(42)
    x = torch.rand(1, 3, 1, 256, 256, generator=g) * 2 - 1
    noise = torch.randn(1, 32, 32, 16, generator=g)
    txt = weights.synth_text_embedding().float()
    legs = {"vae_encode": [], "dit_32_layers": [], "vae_decode": []}

    def chain(record):
        with torch.no_grad():
            t0 = time.perf_counter()
            lat = ref_vae.encode(x).latent
            lat = lat.unsqueeze(2) if lat.dim() == 4 else lat                        # (a one-frame clip comes back as an image) [1, 16, 1, 32, 32]
            t1 = time.perf_counter()
            z = (lat[0].permute(1, 2, 3, 0) - vcfg.shifting_factor) * vcfg.scaling_factor        # infer.py:188
            vid = torch.cat([noise, z, torch.ones_like(z[..., :1])], dim=-1)          # infer.py:54-78 task "sr"
            v = ref_dit(vid=vid.reshape(-1, 33), txt=txt, vid_shape=torch.tensor([[1, 32, 32]]),
                        txt_shape=torch.tensor([[txt.shape[0]]]), timestep=torch.tensor([1000.0])).vid_sample
            x0 = noise - v.reshape(1, 32, 32, 16)                                     # one-step Euler endpoint (euler.py:60-63)
            t2 = time.perf_counter()
            zz = (x0 / vcfg.scaling_factor + vcfg.shifting_factor).permute(3, 0, 1, 2)[None]     # infer.py:236
            ref_vae.decode(zz)
            t3 = time.perf_counter()
        if record:
            legs["vae_encode"].append(t1 - t0); legs["dit_32_layers"].append(t2 - t1); legs["vae_decode"].append(t3 - t2)
        return t3 - t0

    chain(False)
    total = statistics.median([chain(True) for _ in range(3)])
    fv = flops.vae_flops_tiled(vcfg, 1, 256, 256, False)
    f_all = fv["encode"] + fv["decode"] + flops.dit_flops(dcfg, (1, 16, 16))["total"]
    return {"value": 1.0 / total, "unit": "frames/s", "cores": cores, "kind": "reference", "same_workload": True,
            "cpu_tflops": f_all / total / 1e12, "timing": "1 warm-up + median of 3 whole chains",
            "seconds": {"total": total, **{k: statistics.median(v) for k, v in legs.items()}, "model_build": t_build},
            "sample": f"NOT a sample: the identical cfg1 workload (one 256x256 frame through VAE encode -> 32-layer NaDiT-3B -> VAE decode), "
                      f"the reference's own model classes ({rl.kind()}), fp32, PyTorch-SDPA path, {cores} threads"}


def cpu_baseline(flops_per_frame: float) -> dict:
    """The reference's CPU path timed on this host's cores on a bounded sample of the same pipeline: 1 warm-up + median of 3
    per leg, converted to the metric's unit through the algorithmic FLOP ratio (labelled extrapolation).
    ``kind`` "reference": the reference's own NaDiT / VideoAutoencoderKLWrapper classes (PyTorch-SDPA path, fp32), imported by
    oracle/reference_loader.py from the checkout or -- on the GPU box -- from oracle/_ref, the same modules byte-compiled by the
    committed recipe oracle/build_ref.py.  ``kind`` "port" (only when neither is present): the in-repo restatement oracle/*.py,
    calibrated against the reference at 1.06x its time (profiles/r2_cpu_reference_vs_port.json).  Test infrastructure used only
    as a reported baseline."""
    from oracle import dit_oracle, vae_oracle, reference_loader as rl
    config, weights, windows, flops = sub("config"), sub("weights"), sub("windows"), sub("flops")
    torch.set_num_threads(min(torch.get_num_threads(), usable_cores()))     # (never more threads than cores the container may use)
    cores = torch.get_num_threads()
    vcfg = config.VAE_V3
    vsd = {k: v.float() for k, v in weights.synth_vae_state_dict(vcfg).items()}
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3, 5, 96, 96, generator=g) * 2 - 1
    use_ref = rl.available()

    def median3(fn):
        fn()                                            # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    # 2-layer slice (one regular + one shifted window layer) of the 3B-width DiT on a 3x48x48 latent
    dcfg = config.DiTConfig(num_layers=2, mm_layers=1)
    dsd = weights.synth_dit_state_dict(dcfg)
    vid = torch.randn(3, 48, 48, 33, generator=g)
    txt = weights.synth_text_embedding().float()
    if use_ref:
        ref_vae = rl.build_reference_vae(vsd)                       # slicing split 4 + memory limits as configs_3b/main.yaml:53-58
        ref_dit = rl.build_reference_dit(dcfg.as_dict(), {k: v.float() for k, v in dsd.items()})

        def vae_leg():
            with torch.no_grad():
                lat = ref_vae.encode(x[None]).latent
                ref_vae.decode(lat)

        def dit_leg():
            with torch.no_grad():
                ref_dit(vid=vid.reshape(-1, 33), txt=txt, vid_shape=torch.tensor([[3, 48, 48]]),
                        txt_shape=torch.tensor([[txt.shape[0]]]), timestep=torch.tensor([1000.0]))
    else:
        def vae_leg():
            lat = vae_oracle.runner_vae_encode(x, vsd, vcfg)
            vae_oracle.runner_vae_decode(lat, vsd, vcfg)

        def dit_leg():
            dit_oracle.dit_forward(dsd, dcfg, vid, txt, 1000.0, windows_mod=windows)

    t_vae = median3(vae_leg)
    f_vae = sum(flops.vae_flops_tiled(vcfg, 5, 96, 96, False).values())
    t_dit = median3(dit_leg)
    f_dit = flops.dit_flops(dcfg, (3, 24, 24))["total"]
    tflops = (f_vae + f_dit) / (t_vae + t_dit) / 1e12
    what = (f"the reference's own model classes ({rl.kind()}: {'oracle/_ref, byte-compiled by oracle/build_ref.py' if rl.kind() == 'compiled' else rl.REFERENCE_ROOT}), fp32, PyTorch-SDPA path"
            if use_ref else "oracle/*.py (in-repo port of the reference's PyTorch path), fp32")
    # BASELINE config 1 IN FULL (the reference's 32-layer NaDiT-3B + VAE on one 256 x 256 image): tools/cpu_cfg1_full.py, run once per
    # round on a GPU box's host and committed -- building 3.4e9 random fp32 parameters takes longer than this whole measurement
    cfg1_full = None
    for name in ("r6_bench_cfg1.json", "r5_cpu_cfg1_full.json"):
        try:
            one = json.load(open(os.path.join(ROOT, "profiles", name)))
            if "cpu_baseline" in one:                    # a committed `bench.py --workload cfg1` line: GPU and CPU legs of the SAME workload
                cb = one["cpu_baseline"]
                cfg1_full = {"frames_per_s_cfg1": cb["value"], "cpu_tflops": cb["cpu_tflops"], "cores": cb["cores"], "seconds": cb["seconds"],
                             "gpu_frames_per_s_cfg1": one["value"],
                             "source": f"profiles/{name} (bench.py --workload cfg1: both legs on one box in one run; NOT timed in this run)"}
            else:
                cfg1_full = {"frames_per_s_cfg1": one["frames_per_s_cfg1"], "cpu_tflops": one["cpu_tflops"], "cores": one["cores"],
                             "seconds": one["seconds"], "source": f"profiles/{name} (tools/cpu_cfg1_full.py on a GPU box's host; NOT timed in this run)"}
            break
        except (OSError, KeyError, ValueError):
            continue
    return {"value": tflops * 1e12 / flops_per_frame, "unit": "frames/s", "cores": cores, "kind": "reference" if use_ref else "port",
            "cpu_tflops": tflops, "timing": "1 warm-up + median of 3 per leg", "cfg1_full": cfg1_full,
            "sample": f"{what}: full VAE enc+dec of a 5x96x96 clip ({t_vae:.2f}s) + 2-layer (regular + shifted windows) "
                      f"3B-width DiT on a 3x48x48 latent ({t_dit:.2f}s); extrapolated to the workload by algorithmic FLOPs"}


def respawn_under_torchrun(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks of this script on this node (one per GPU, RCCL)."""
    if "--cpu-double" not in sys.argv and (not torch.cuda.is_available() or torch.cuda.device_count() < n):
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        print(f"[bench] --gpus {n} requested but only {have} GPU(s) are visible: refusing to report a {n}-GPU number",
              file=sys.stderr)
        return 2
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (the host driver supports nothing else)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print per-phase timings to stderr")
    ap.add_argument("--two-step-upsampler", action="store_true",
                    help="A/B: run the VAE's spatial upsampler as the reference's upscale_conv + pixel shuffle + conv instead of its sub-pixel form")
    ap.add_argument("--three-tap-head", action="store_true",
                    help="A/B: keep three temporal taps on the first frame of every clip instead of the two-term sum of the taps "
                         "that fall on the replicated frame (VideoVAEEngine(merge_causal_head=False))")
    ap.add_argument("--bf16-trunk", action="store_true",
                    help="A/B: round 2's storage regime -- the VAE's residual trunk and the DiT's residual stream in bf16 instead of "
                         "h16 / fp32 (48.2 instead of 50 dB end to end against the fp32 reference at production width)")
    ap.add_argument("--trunk", choices=["h16", "fp32", "bf16"], default=None,
                    help="A/B: storage of the VAE's residual trunk (VideoVAEEngine(trunk_store=...)); default: the engine's (h16; round 3: fp32)")
    ap.add_argument("--stream", choices=["h16", "fp32", "bf16"], default=None,
                    help="A/B: storage of the NaDiT's residual stream (NaDiTEngine(hid_store=...)); default: the engine's")
    ap.add_argument("--tile-streams", type=int, default=None,
                    help="A/B: HIP streams the VAE's spatial tiles are issued on (VideoVAEEngine(tile_streams=...); default: the "
                         "engine's, 1 = every launch on one stream)")
    ap.add_argument("--branch", choices=["h16", "fp32", "bf16"], default=None,
                    help="A/B: storage of conv1's output inside a VAE block (VideoVAEEngine(branch_store=...)); default: the engine's (h16; round 3: bf16)")
    ap.add_argument("--by-shape", default=None, metavar="FILE",
                    help="also write the per-(kernel class, shape) table of the timed steps' GEMM / conv launches (launches, avg us, "
                         "TFLOP/s, ms per step) to FILE")
    ap.add_argument("--cpu-double", action="store_true",
                    help="TESTS ONLY (needs SVR_BENCH_ALLOW_CPU_DOUBLE=1): CPU ranks over gloo with the torch double of the C ABI and "
                         "reduced-width models -- exercises this script's launcher / gather / guard / JSON plumbing, measures nothing")
    args = ap.parse_args()
    double = args.cpu_double
    if double and os.environ.get("SVR_BENCH_ALLOW_CPU_DOUBLE") != "1":
        print("[bench] --cpu-double is a test mode (tests/test_bench_launcher.py); the product path needs the HIP library and a GPU",
              file=sys.stderr)
        sys.exit(2)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(respawn_under_torchrun(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:       # (checked before the rendezvous, which would wait for peers)
        print(f"[bench] launched with WORLD_SIZE={os.environ.get('WORLD_SIZE')} but --gpus {args.gpus}: the two must agree",
              file=sys.stderr)
        sys.exit(2)
    dist_mod = sub("dist")
    rank, world, local = dist_mod.init_from_env(backend="gloo" if double else None)
    if not double and local >= torch.cuda.device_count():
        print(f"[bench] rank {rank}: LOCAL_RANK {local} has no GPU ({torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    device = torch.device("cpu" if double else f"cuda:{local}")
    if not double:
        torch.cuda.set_device(device)
    Tick = _HostTick if double else torch.cuda.Event
    sync = (lambda: None) if double else torch.cuda.synchronize

    config, weights, flops = sub("config"), sub("weights"), sub("flops")
    frames, H, W, tiled, desc = WORKLOADS[args.workload]
    ops = make_double_ops() if double else make_profiled_ops(device)
    dcfg, vcfg = (config.DIT_7B if args.workload == "cfg5" else config.DIT_3B), config.VAE_V3
    if double:                                            # reduced width, same graphs: the plumbing is what runs here
        dcfg, vcfg = config.DIT_TINY, config.VAE_TINY
        if args.workload == "cfg4":
            frames = 10
            CFG4.update(in_hw=(24, 40), resolution=48, batch_size=5, temporal_overlap=1)
        else:
            frames, H, W = min(frames, 5), min(H, 32), min(W, 48)
    # random-init weights of the exact architecture, generated on the GPU (no checkpoints available offline)
    dit = sub("dit").NaDiTEngine(dcfg, weights.synth_dit_state_dict(dcfg, device=device), ops,
                                 hid_store="bf16" if args.bf16_trunk else args.stream)
    vae = sub("vae").VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg, device=device), ops,
                                    merge_upsamplers=not args.two_step_upsampler, merge_causal_head=not args.three_tap_head,
                                    trunk_store="bf16" if args.bf16_trunk else args.trunk, branch_store=args.branch,
                                    **({} if args.tile_streams is None else {"tile_streams": args.tile_streams}))
    runner_mod = sub("runner")
    tile, tile_ov = ((32, 32), (8, 8)) if double else ((1024, 1024), (128, 128))
    runner = runner_mod.VideoDiffusionInfer(
        runner_mod.default_config(dcfg, vcfg), encode_tiled=tiled, encode_tile_size=tile, encode_tile_overlap=tile_ov,
        decode_tiled=tiled, decode_tile_size=tile, decode_tile_overlap=tile_ov)
    runner.dit, runner.vae = dit, vae
    runner.configure_diffusion(device=device, dtype=torch.bfloat16)

    sharded = args.workload == "cfg4"
    g = torch.Generator(device=device).manual_seed(42 + (0 if sharded else rank))
    Tl, hl, wl = (frames - 1) // 4 + 1, H // 8, W // 8
    txt = weights.synth_text_embedding(device=device)
    phase = {"encode": 0.0, "dit": 0.0, "decode": 0.0, "gather": 0.0}
    if sharded:
        # the same synthetic clip on every rank (each rank reads the frames of the batches it owns)
        images = torch.rand(frames, CFG4["in_hw"][0], CFG4["in_hw"][1], 3, generator=g, device=device)
        pipe_kw = {k: v for k, v in CFG4.items() if k != "in_hw"}
        plans, _ = sub("pipeline").plan_batches(frames, CFG4["batch_size"], CFG4["temporal_overlap"], CFG4["uniform_batch_size"])

        def step(timed: bool):
            out = dist_mod.upscale_sharded(images, runner, txt, **pipe_kw)      # [128, 2160, 3840, 3] on every rank
            return out, None
    else:
        useful = frames - 1 if frames > 1 and frames % 4 == 1 else frames      # frames of the caller's clip in one padded batch
        x = (torch.rand(3, frames, H, W, generator=g, device=device) * 2 - 1).to(torch.bfloat16)
        noise = torch.randn(Tl, hl, wl, 16, generator=g, device=device).to(torch.bfloat16)

        def step(timed: bool):
            ev = [Tick(enable_timing=True) for _ in range(5)] if timed else None
            if timed: ev[0].record()
            lat = runner.vae_encode([x])[0]
            if timed: ev[1].record()
            cond = runner.get_condition(noise, latent_blur=lat, task="sr")
            x0 = runner.inference([noise], [cond], [txt], [txt])[0]
            if timed: ev[2].record()
            # the 4n+1 rule pads the clip with one reversed frame (generation_phases.py:398-404) that the pipeline trims after
            # decode; as in pipeline.upscale (skip_trimmed_frames), the causal decoder is told not to produce it
            out = runner.vae_decode([x0], keep_frames=[useful])[0]     # [3, useful, H, W] view of THWC
            if timed: ev[3].record()
            thwc = out.permute(1, 2, 3, 0) if out.dim() == 4 else out.permute(1, 2, 0)[None]
            gathered = dist_mod.all_gather_frames(thwc.contiguous())
            if timed: ev[4].record()
            return gathered, ev

    def fingerprint(t):
        """fp64 sum and sum of squares of an output (on the device, OUTSIDE the timed region).  Accumulated over chunks of
        eight frames in a fixed order: the gathered clip of an 8-rank run is 12.7 GB of bf16, and two fp64 images of it next
        to the engines' cached activations would not be a safe assumption even on 288 GB."""
        acc = torch.zeros(2, dtype=torch.float64, device=t.device)
        for c in (t.split(8, 0) if t.dim() > 1 else (t,)):
            d = c.double()
            acc[0] += d.sum()
            acc[1] += (d * d).sum()
        return acc

    first_print = None
    for _ in range(args.warmup):
        out, _ = step(False)
        if first_print is None:
            first_print = fingerprint(out)
        del out
    sync()
    # the matrix-pipe rate this device sustains under its power limit, measured HERE (svr_mfma_calibrate: a bare MFMA loop on random
    # operands, ~0.25 s) right before and right after the timed region -- on every rank at the same time, so that node-level power /
    # thermal coupling between the GPUs of one node shows up as a lower calibrated peak than the 1-GPU run's
    if world > 1:
        torch.distributed.barrier()
    cal = [None, None]
    if not double:
        cal[0] = ops.mfma_calibrate()
    if world > 1:
        torch.distributed.barrier()
    ops.recording = True
    evs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last_out, ev = step(True)
        evs.append(ev)
    sync()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    ops.recording = False
    if not double:
        cal[1] = ops.mfma_calibrate()
    # ---- output guards (after the timed region): a kernel that exits early, or a launch that was dropped, must not post a record.
    # (1) every value of the last step's output is finite; (2) its statistics sit in the band the REFERENCE's decoder produces
    # with these synthetic weights (tests/golden/vae_tile1024.pt records mean / std / min / max of the reference's fp32 decode
    # at the real tile size: std 0.58, range about [-3.5, 3]); for the [0, 1] frames of the cfg4 pipeline the band is that of
    # clamped frames; (3) the hot path is deterministic, so the last timed step must reproduce the first warm-up step's
    # output sums bit for bit (same inputs).  On failure: no JSON line, exit code 3.
    guard = {"finite": all(bool(torch.isfinite(c).all()) for c in (last_out.split(8, 0) if last_out.dim() > 1 else (last_out,)))}
    fp_last = fingerprint(last_out)
    n_el = last_out.numel()
    g_mean, g_std = float(fp_last[0]) / n_el, math.sqrt(max(float(fp_last[1]) / n_el - (float(fp_last[0]) / n_el) ** 2, 0.0))
    try:
        gold = torch.load(os.path.join(ROOT, "tests", "golden", "vae_tile1024.pt"), weights_only=True)
        ref_mean, ref_std = float(gold["dec_mean"]), float(gold["dec_std"])
    except (OSError, KeyError):
        ref_mean, ref_std = 0.0, 0.58
    if double:
        band_ok = True                                       # (reduced-width stand-in models: the band of the real decoder does not apply)
    elif sharded:
        band_ok = 0.05 <= g_std <= 0.6 and 0.2 <= g_mean <= 0.8          # [0, 1] frames after clamp and colour fix
    else:
        band_ok = 0.5 * ref_std <= g_std <= 2.0 * ref_std and abs(g_mean - ref_mean) <= ref_std
    guard.update(mean=g_mean, std=g_std, band_ok=band_ok, reference_decode_mean=ref_mean, reference_decode_std=ref_std,
                 deterministic=None if first_print is None else bool(torch.equal(first_print, fp_last)))
    bad = (not guard["finite"]) or (not band_ok) or guard["deterministic"] is False
    if world > 1:
        flag = torch.tensor([int(bad)], device=device)
        torch.distributed.all_reduce(flag)
        bad = bool(flag.item())
    if bad:
        print(f"[bench] rank {rank}: output guard FAILED, refusing to report a number: {guard}", file=sys.stderr)
        if world > 1:
            torch.distributed.destroy_process_group()
        sys.exit(3)
    del last_out
    n_ranks = 1
    own_dt = dt
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)
        cnt = torch.ones(1, device=device, dtype=torch.int64)
        torch.distributed.all_reduce(cnt)                       # ranks that actually ran the timed region
        n_ranks = int(cnt)
    if not sharded:
        for ev in evs:
            for i, k in enumerate(("encode", "dit", "decode", "gather")):
                phase[k] += ev[i].elapsed_time(ev[i + 1])
        phase = {k: v / args.steps for k, v in phase.items()}
    # per-rank view of the same run (a multi-rank line that only carries the max-over-ranks wall time cannot show a straggler or a
    # node that clocks all its GPUs down): every rank's phase times, own wall time and calibrated MFMA rate, gathered after the timing
    own = [own_dt / args.steps * 1e3, phase["encode"], phase["dit"], phase["decode"], phase["gather"], cal[0] or 0.0, cal[1] or 0.0]
    per_rank = [own]
    if world > 1:
        mine = torch.tensor(own, device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = [[float(v) for v in t.tolist()] for t in allr]

    if rank == 0:
        if sharded:        # 8 temporal batches of 17 frames (latent 5 x 270 x 480), tiled VAE
            bf = CFG4["batch_size"]
            f_dit = flops.dit_flops(dcfg, ((bf - 1) // 4 + 1, hl // 2, wl // 2))
            f_vae = flops.vae_flops_tiled(vcfg, bf, H, W, tiled)
            f_step = len(plans) * (f_dit["total"] + f_vae["encode"] + f_vae["decode"])
            f_exec = len(plans) * (f_dit["total"] + sum(flops.vae_flops_tiled(vcfg, bf, H, W, tiled, merged_upsamplers=not args.two_step_upsampler, causal_head=not args.three_tap_head).values()))
            frames_per_step, padded_per_step = frames, frames
        else:
            f_dit = flops.dit_flops(dcfg, (Tl, hl // 2, wl // 2))
            f_vae = flops.vae_flops_tiled(vcfg, frames, H, W, tiled)
            f_step = f_dit["total"] + f_vae["encode"] + f_vae["decode"]
            f_exec = f_dit["total"] + sum(flops.vae_flops_tiled(vcfg, frames, H, W, tiled, merged_upsamplers=not args.two_step_upsampler,
                                                                causal_head=not args.three_tap_head, keep_frames=useful).values())
            # `value` counts the frames of the caller's clip (32 of the 33-frame padded batch: the padding frame is encoded and goes
            # through the DiT, the decoder skips it); `padded_frames_per_s` is the round-1/2 convention (all 33)
            frames_per_step = world * useful
            padded_per_step = world * frames
        kern = ops.summary()
        if args.by_shape and not double:
            with open(args.by_shape, "w") as fh:
                fh.write(ops.by_shape(args.steps))
        # dominant kernel: the LDS-halo implicit-GEMM conv (63 % of the step, profiles/r3_cfg3_kernel_stats.csv) -- its launches
        # ALONE (the thin-output / thin-input / sub-pixel / generic conv kernels are separate classes of `per_kernel`)
        dom = kern.get("conv_halo") or kern.get("conv_generic") or kern.get("gemm_persistent") or kern["gemm"]
        c_flops, c_sec, c_n = dom["flops"], dom["seconds"], dom["launches"]
        traffic, traffic_note = None, None
        shader_clock = None
        for name in ("r6_cfg3_pmc_traffic.json", "r5_cfg3_pmc_traffic.json", "r4_cfg3_pmc_traffic.json", "r3_cfg3_pmc_traffic.json", "r2_cfg3_pmc_traffic.json",
                     "r1_cfg3_pmc_traffic.json"):
            # HBM bytes per launch come from separate rocprofv3 --pmc passes of this workload (counters cannot be collected
            # inside the timed run); the file names the commit it was measured on
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                if pmc.get("workload") == "cfg3" and args.workload == "cfg3" and "conv_halo" in kern:
                    krec = pmc["kernels"][pmc.get("dominant", "svr::conv_halo2_kernel")]
                    traffic = krec["hbm_bytes_per_launch"]
                    shader_clock = krec.get("shader_clock_ghz")
                    traffic_note = f"bytes/launch, FETCH_SIZE*2 + WRITE_SIZE from profiles/{name}" + \
                                   (f" (kernels as of {pmc['measured_at']})" if pmc.get("measured_at") else "")
                    break
            except (OSError, KeyError, ValueError):
                continue
        roof = {"bound": "mfma", "kernel": "svr::conv_halo2_kernel<16, 3> (LDS-halo implicit-GEMM causal Conv3d, register-streamed weights, 16x32-voxel patches x 128 couts, 8 rows x 64 couts per wave, 3x3 spatial taps)",
                "achieved": c_flops / max(c_sec, 1e-12) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": c_flops / max(c_sec, 1e-12) / 1e12 / PEAK_BF16_TFLOPS, "traffic": traffic,
                "traffic_note": traffic_note,
                # effective shader clock of this kernel under load = GRBM_GUI_ACTIVE cycles / launch duration, from the same PMC
                # passes (the chip is power-managed: nominal 2.4 GHz is what `peak` assumes)
                "shader_clock_ghz": shader_clock,
                "timing_note": ("HIP events on the launch stream; with tile_streams > 1 another tile's kernels share the chip while "
                                "a launch runs, so durations include that overlap (--tile-streams 1 isolates the kernel)")
                               if getattr(vae, "tile_streams", 1) > 1 else "HIP events on the launch stream",
                # in-run calibration (svr_mfma_calibrate right before / after the timed region on this device): the bare-MFMA rate
                # the power limit allows on random operands; `frac_of_power_limited` separates "this box clocks lower" from "the
                # kernel got worse" inside the record itself (the nominal `frac` above cannot)
                "power_limited_peak": None if cal[0] is None else {
                    "tflops_before": cal[0], "tflops_after": cal[1], "tflops": min(cal), "unit": "TFLOP/s",
                    "what": "bare v_mfma_f32_32x32x16_bf16 loop on random bf16 operands, 4 x 512 threads per CU, ~0.25 s, median of 3 "
                            "(svr_mfma_calibrate), on this device in this run"},
                "frac_of_power_limited": None if cal[0] is None else c_flops / max(c_sec, 1e-12) / 1e12 / min(cal),
                "launches": c_n, "avg_launch_us": c_sec / max(c_n, 1) * 1e6,
                "algorithmic_flops_per_launch": c_flops / max(c_n, 1),
                "share_of_step_time": c_sec / max(dt, 1e-12),
                "per_kernel": {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "tflops": round(v["tflops"], 1)}
                               for k, v in kern.items()}}
        family = "7B" if args.workload == "cfg5" else "3B"
        res = {
            "metric": "upscaled frames/sec (720p->4K, SeedVR2-3B)" if args.workload in ("cfg3", "cfg4")
                      else f"upscaled frames/sec ({args.workload}, SeedVR2-{family})",
            "value": frames_per_step * args.steps / dt, "unit": "frames/s", "n_gpus": n_ranks, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "storage": f"bf16 activations (every MFMA operand); VAE residual trunk {vae.trunk_store}, conv1 outputs {vae.branch_store} "
                       f"(h16 = IEEE half of x * 2^-6); DiT residual stream {dit.hid_store}",
            "config": {"workload": f"{args.workload}: {desc}",
                       "frames_per_step_per_gpu": useful if not sharded else f"{frames} per clip, 8 batches of 17 shared by the ranks",
                       "pixels": [H, W], "latent": [Tl, hl, wl] if not sharded else [(CFG4["batch_size"] - 1) // 4 + 1, hl, wl],
                       "vae_tiled": tiled, "parallelism": f"dp{world}",
                       "weights": f"random-init SeedVR2-{family} + video_vae_v3 architecture (seeded)"},
            "padded_frames_per_s": padded_per_step * args.steps / dt,
            # FLOPs of the reference's algorithm for this workload, and of what this engine executes for it (the spatial-only
            # VAE upsampler runs in its sub-pixel form: same function, fewer multiply-adds); the achieved rate counts the latter
            "algorithmic_tflop_per_step": f_step / 1e12,
            "executed_tflop_per_step": f_exec / 1e12,
            "achieved_tflops_per_gpu": f_exec * (1 if sharded else world) * args.steps / dt / 1e12 / world,
            "roofline": roof,
            "vae_tile_streams": getattr(vae, "tile_streams", 1),
            "output_guard": guard,
        }
        # DESIGN.md section 6's prediction for THIS launch, made falsifiable: the compute term is A PRIORI -- the committed 1-GPU line of
        # the same workload (profiles/), never this run's own timings -- plus the communication model (direct xGMI transfers at
        # ~64 GB/s per direction and link).  What this run measured is reported next to it, with the difference.
        if world > 1 or sharded:
            frame_bytes = H * W * 3 * (4 if sharded else 2)      # (the pipeline's frames are fp32, the runner's decode output bf16)
            one_gpu_s, src = None, None
            for name in ((f"r6_bench_{args.workload}_1gpu.json", f"r5_bench_{args.workload}_1gpu.json", f"r4_bench_{args.workload}_1gpu.json")
                         if sharded else (f"r6_bench_{args.workload}.json", f"r5_bench_{args.workload}.json")):
                try:
                    one = json.load(open(os.path.join(ROOT, "profiles", name)))
                    one_gpu_s, src = float(one["ms_per_step"]) / 1e3, f"profiles/{name}"
                    break
                except (OSError, KeyError, ValueError):
                    continue
            if sharded:
                n_b = len(plans)
                per_batch_s = (one_gpu_s if one_gpu_s is not None else 41.2) / 8
                comm = (frames * frame_bytes / max(world, 1)) / 64e9 + 0.001 * (n_b - 1)
                pred = math.ceil(n_b / world) * per_batch_s + (comm if world > 1 else 0.0)
                model = (f"ceil({n_b} batches / {world} ranks) x {per_batch_s:.2f} s per batch (the committed 1-GPU line {src}: "
                         f"a priori) + gather of the clip over xGMI")
            else:
                comm = useful * frame_bytes / 64e9 + 0.003
                pred = None if one_gpu_s is None else one_gpu_s + (comm if world > 1 else 0.0)
                model = (f"the committed 1-GPU step of this workload ({src}: a priori, not this run's timings) + all-gather: each link "
                         "carries one rank's frames once at 64 GB/s")
            res["predicted_s"] = {"per_step": pred, "model": model, "comm_s_model": comm if world > 1 else 0.0,
                                  "measured_per_step": dt / args.steps,
                                  "measured_minus_predicted_s": None if pred is None else dt / args.steps - pred,
                                  "own_run_compute_s_rank0": None if sharded else (phase["encode"] + phase["dit"] + phase["decode"]) / 1e3}
        if world > 1:
            cols = ("step_ms", "encode_ms", "dit_ms", "decode_ms", "gather_ms", "mfma_calibrated_tflops_before", "mfma_calibrated_tflops_after")
            stats = {}
            for j, cname in enumerate(cols):
                col = [r[j] for r in per_rank]
                stats[cname] = {"min": min(col), "median": statistics.median(col), "max": max(col),
                                "argmax_rank": col.index(max(col)), "argmin_rank": col.index(min(col))}
            res["ranks"] = {"n": len(per_rank), "stats": stats,
                            "note": "per-rank wall time per step, phase times (HIP events on each rank's launch stream) and the bare-MFMA rate "
                                    "each device sustained right before / after the timed region, all ranks calibrating at the same time: a "
                                    "straggler shows as argmax_rank of step_ms, node-level power coupling as calibrated rates below the 1-GPU "
                                    "line's roofline.power_limited_peak"}
        if not sharded:
            dit_tf = f_dit["total"] / max(phase["dit"], 1e-9) / 1e9
            res.update({"dit_ms_per_step": phase["dit"], "vae_encode_ms": phase["encode"], "vae_decode_ms": phase["decode"],
                        "allgather_ms": phase["gather"], "dit_tflops": dit_tf,
                        # whole-DiT-step MFMA fraction: algorithmic FLOPs of the forward / its measured time / dense bf16 peak
                        "dit_mfma_frac": dit_tf / PEAK_BF16_TFLOPS})
        if double:
            res.update(test_mode=True, data="synthetic; TEST MODE (--cpu-double): torch double of the C ABI on CPU ranks over gloo, "
                                            "reduced-width models -- plumbing only, the numbers mean nothing")
        if not args.no_cpu_baseline and world == 1 and not double:      # (the CPU leg is a 1-GPU line item: the other ranks would idle at the barrier)
            res["cpu_baseline"] = (cpu_baseline_cfg1() if args.workload == "cfg1" else
                                   cpu_baseline(f_step / frames_per_step * (world if not sharded else 1)))
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
