#!/usr/bin/env python
"""Command-line surface of the reference (inference_cli.py:1346-1481, 45 flags) in front of the MI355X hot path.

    python inference_cli.py clip.mp4 --resolution 2160 --batch_size 33 --uniform_batch_size --temporal_overlap 3 \\
           --vae_encode_tiled --vae_decode_tiled --cuda_device 0,1,2,3,4,5,6,7

Every flag of the reference parses with the same name, type, default and choices (tests/test_api_surface.py compares the
two parsers by ``ast``), so existing scripts keep working.  What the flags DO is deliberately thin:
  * frames -> ``pipeline.upscale`` (one GPU) or ``dist.upscale_sharded`` (--cuda_device a,b,...: one process per GPU over
    RCCL, temporal batches dealt round-robin; replaces the reference's mp.Process + mp.Queue workers,
    inference_cli.py:1127-1288);
  * flags that select a small-VRAM policy (--blocks_to_swap, --swap_io_components, --*_offload_device, --cache_dit/_vae,
    --chunk_size), another attention backend or torch.compile are accepted and have no effect: one attention kernel, no
    tracing compiler, 288 GB of HBM;
  * media I/O is plumbing, not the hot path: images through PIL, tensors as .pt / .npy, video through OpenCV when it is
    installed (the reference's own dependency).
"""
import argparse
import os
import platform
import subprocess
import sys
import time
from typing import List, Optional

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "comfyui-seedvr2_videoupscaler_amd"
VIDEO_EXT = {".mp4", ".avi", ".mov", ".mkv", ".webm", ".m4v"}
IMAGE_EXT = {".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp"}
TENSOR_EXT = {".pt", ".npy"}


def _itf():
    import importlib
    return importlib.import_module(f"{PKG}.interfaces")


def build_parser() -> argparse.ArgumentParser:
    itf = _itf()
    parser = argparse.ArgumentParser(
        description="SeedVR2 Video Upscaler - CLI for high-quality image/video upscaling and batch processing (MI355X backend)",
        formatter_class=argparse.RawDescriptionHelpFormatter, allow_abbrev=False)
    g = parser.add_argument_group("Input/Output options")
    g.add_argument("input", type=str, help="Input: video file, image file, tensor file (.pt / .npy, [T, H, W, 3] in [0, 1]) or directory")
    g.add_argument("--output", type=str, default=None, help="Output path (default: auto-generated in 'output/' directory)")
    g.add_argument("--output_format", type=str, default=None, choices=["mp4", "png", None],
                   help="Output format: 'mp4' (video) or 'png' (image sequence). Default: auto-detect from input type")
    g.add_argument("--video_backend", type=str, default="opencv", choices=["opencv", "ffmpeg"], help="Video encoder backend")
    g.add_argument("--10bit", dest="use_10bit", action="store_true", help="Save 10-bit video with x265 (requires --video_backend ffmpeg)")
    g.add_argument("--model_dir", type=str, default=None, help="Model directory (default: ./models/SEEDVR2)")
    g = parser.add_argument_group("Model selection")
    g.add_argument("--dit_model", type=str, default=itf.DEFAULT_DIT, choices=list(itf.DIT_MODELS), help="DiT model to use (3B / 7B, fp16 / fp8)")
    g = parser.add_argument_group("Processing parameters")
    g.add_argument("--resolution", type=int, default=1080, help="Target short-side resolution in pixels (default: 1080)")
    g.add_argument("--max_resolution", type=int, default=0, help="Maximum resolution for any edge. 0 = no limit (default: 0)")
    g.add_argument("--batch_size", type=int, default=5, help="Frames per batch (4n+1: 1, 5, 9, 13, 17, 21, ...). Default: 5")
    g.add_argument("--uniform_batch_size", action="store_true", help="Pad final batch to match batch_size")
    g.add_argument("--seed", type=int, default=42, help="Random seed for reproducibility (default: 42)")
    g.add_argument("--skip_first_frames", type=int, default=0, help="Skip N initial frames (default: 0)")
    g.add_argument("--load_cap", type=int, default=0, help="Load maximum N frames from video. 0 = load all (default: 0)")
    g.add_argument("--chunk_size", type=int, default=0, help="Frames per chunk for streaming mode (accepted; clips stay resident in HBM)")
    g.add_argument("--prepend_frames", type=int, default=0, help="Prepend N reversed frames to reduce start artifacts (auto-removed). Default: 0")
    g.add_argument("--temporal_overlap", type=int, default=0, help="Frames to overlap between batches/GPUs for smooth blending (default: 0)")
    g = parser.add_argument_group("Quality control")
    g.add_argument("--color_correction", type=str, default="lab", choices=list(itf.COLOR_CORRECTIONS), help="Color correction method (default: lab)")
    g.add_argument("--input_noise_scale", type=float, default=0.0, help="Input noise injection scale (0.0-1.0) (default: 0.0)")
    g.add_argument("--latent_noise_scale", type=float, default=0.0, help="Latent noise injection scale (0.0-1.0) (default: 0.0)")
    g = parser.add_argument_group("Device management")
    if platform.system() != "Darwin":
        g.add_argument("--cuda_device", type=str, default=None, help="GPU(s): single '0' or multi-GPU '0,1,2'. Default: device 0")
    g.add_argument("--dit_offload_device", type=str, default="none", help="accepted, no effect (models stay resident in HBM)")
    g.add_argument("--vae_offload_device", type=str, default="none", help="accepted, no effect")
    g.add_argument("--tensor_offload_device", type=str, default="cpu", help="accepted, no effect (intermediates stay in HBM)")
    g = parser.add_argument_group("Memory optimization (BlockSwap)")
    g.add_argument("--blocks_to_swap", type=int, default=0, help="accepted, no effect")
    g.add_argument("--swap_io_components", action="store_true", help="accepted, no effect")
    g = parser.add_argument_group("VAE tiling (for high resolution upscale)")
    g.add_argument("--vae_encode_tiled", action="store_true", help="Enable VAE encode tiling")
    g.add_argument("--vae_encode_tile_size", type=int, default=1024, help="VAE encode tile size in pixels (default: 1024)")
    g.add_argument("--vae_encode_tile_overlap", type=int, default=128, help="VAE encode tile overlap in pixels (default: 128)")
    g.add_argument("--vae_decode_tiled", action="store_true", help="Enable VAE decode tiling")
    g.add_argument("--vae_decode_tile_size", type=int, default=1024, help="VAE decode tile size in pixels (default: 1024)")
    g.add_argument("--vae_decode_tile_overlap", type=int, default=128, help="VAE decode tile overlap in pixels (default: 128)")
    g.add_argument("--tile_debug", type=str, default="false", choices=["false", "encode", "decode"], help="accepted, no effect")
    g = parser.add_argument_group("Performance optimization")
    g.add_argument("--attention_mode", type=str, default="sdpa", choices=list(itf.ATTENTION_MODES),
                   help="accepted; every mode runs the hand-written HIP window-attention kernel (results of the SDPA path)")
    g.add_argument("--compile_dit", action="store_true", help="accepted, no effect (no tracing compiler on the HIP path)")
    g.add_argument("--compile_vae", action="store_true", help="accepted, no effect")
    g.add_argument("--compile_backend", type=str, default="inductor", choices=["inductor", "cudagraphs"], help="accepted, no effect")
    g.add_argument("--compile_mode", type=str, default="default",
                   choices=["default", "reduce-overhead", "max-autotune", "max-autotune-no-cudagraphs"], help="accepted, no effect")
    g.add_argument("--compile_fullgraph", action="store_true", help="accepted, no effect")
    g.add_argument("--compile_dynamic", action="store_true", help="accepted, no effect")
    g.add_argument("--compile_dynamo_cache_size_limit", type=int, default=64, help="accepted, no effect")
    g.add_argument("--compile_dynamo_recompile_limit", type=int, default=128, help="accepted, no effect")
    g = parser.add_argument_group("Model caching (batch processing)")
    g.add_argument("--cache_dit", action="store_true", help="accepted (engines always stay resident within a process)")
    g.add_argument("--cache_vae", action="store_true", help="accepted")
    g = parser.add_argument_group("Debugging")
    g.add_argument("--debug", action="store_true", help="Enable verbose logging")
    return parser


# ---------------------------------------------------------------------------------------------------------
def load_frames(path: str, skip: int = 0, cap: int = 0):
    """-> (frames [T, H, W, 3] float32 in [0, 1], fps)."""
    import numpy as np
    import torch
    ext = os.path.splitext(path)[1].lower()
    fps = 30.0
    if os.path.isdir(path):
        raise ValueError(f"{path} is a directory: main() runs each media file in it as its own job (list_inputs)")
    elif ext in TENSOR_EXT:
        t = torch.load(path, weights_only=True) if ext == ".pt" else torch.from_numpy(np.load(path))
        frames = t.float()
        if frames.dim() == 3:
            frames = frames[None]
    elif ext in IMAGE_EXT:
        from PIL import Image
        frames = torch.from_numpy(np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0)[None]
    elif ext in VIDEO_EXT:
        try:
            import cv2  # type: ignore
        except ImportError as e:
            raise RuntimeError("reading video files needs OpenCV (the reference's own dependency); pass an image folder or a .pt tensor") from e
        cap_ = cv2.VideoCapture(path)
        fps = cap_.get(cv2.CAP_PROP_FPS) or 30.0
        out = []
        while True:
            ok, f = cap_.read()
            if not ok:
                break
            out.append(torch.from_numpy(cv2.cvtColor(f, cv2.COLOR_BGR2RGB)).float() / 255.0)
        cap_.release()
        frames = torch.stack(out)
    else:
        raise ValueError(f"unsupported input: {path}")
    if skip > 0:
        frames = frames[skip:]
    if cap > 0:
        frames = frames[:cap]
    if frames.shape[0] == 0:
        raise ValueError("No frames to process")
    return frames[..., :3].contiguous(), fps


def save_frames(frames, path: str, fmt: str, fps: float = 30.0):
    import numpy as np
    import torch
    arr = (frames.float().clamp(0, 1) * 255.0).round().to(torch.uint8).cpu().numpy()
    if fmt == "pt":
        torch.save(frames.cpu(), path)
    elif fmt == "png":
        from PIL import Image
        if arr.shape[0] == 1 and path.lower().endswith(".png"):
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            Image.fromarray(arr[0]).save(path)
        else:
            os.makedirs(path, exist_ok=True)
            for i, a in enumerate(arr):
                Image.fromarray(a).save(os.path.join(path, f"frame_{i:06d}.png"))
    else:
        import cv2  # type: ignore
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        w = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (arr.shape[2], arr.shape[1]))
        for a in arr:
            w.write(np.ascontiguousarray(a[..., ::-1]))
        w.release()


def list_inputs(path: str) -> List[str]:
    """A directory input = one job per media file in it (images AND videos, sorted by name), as the reference's CLI processes
    a folder (inference_cli.py: get_media_files / process_single_file): files of different sizes never meet in one clip and
    unrelated images are not blended through the temporal overlap.  A file input = that one job."""
    if not os.path.isdir(path):
        return [path]
    files = sorted(os.path.join(path, f) for f in os.listdir(path)
                   if os.path.splitext(f)[1].lower() in IMAGE_EXT | VIDEO_EXT | TENSOR_EXT)
    if not files:
        raise ValueError(f"no images, videos or tensors in {path}")
    return files


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def default_output(inp: str, fmt: str) -> str:
    stem = os.path.splitext(os.path.basename(os.path.normpath(inp)))[0]
    return os.path.join("output", f"{stem}_upscaled" + ("" if fmt == "png_dir" else f".{fmt}"))


def run(args, frames):
    """One rank's work: engines resident on cuda:LOCAL_RANK, single-GPU or sharded pipeline; returns the full clip."""
    import importlib
    import torch
    itf = _itf()
    dist_mod = importlib.import_module(f"{PKG}.dist")
    pipeline = importlib.import_module(f"{PKG}.pipeline")
    rank, world, local = dist_mod.init_from_env()
    device = f"cuda:{local}"
    vae_cfg = dict(model=itf.DEFAULT_VAE, device=device, encode_tiled=args.vae_encode_tiled,
                   encode_tile_size=args.vae_encode_tile_size, encode_tile_overlap=args.vae_encode_tile_overlap,
                   decode_tiled=args.vae_decode_tiled, decode_tile_size=args.vae_decode_tile_size,
                   decode_tile_overlap=args.vae_decode_tile_overlap)
    runner = itf.get_runner(dict(model=args.dit_model, device=device), vae_cfg, args.model_dir)
    text = itf.load_text_embedding(runner.dit.device, args.model_dir)
    kw = dict(resolution=args.resolution, max_resolution=args.max_resolution, batch_size=args.batch_size,
              uniform_batch_size=args.uniform_batch_size, temporal_overlap=args.temporal_overlap,
              prepend_frames=args.prepend_frames, color_correction=args.color_correction,
              input_noise_scale=args.input_noise_scale, latent_noise_scale=args.latent_noise_scale, seed=args.seed)
    t0 = time.time()
    if world > 1:                                      # only rank 0 writes the result: gather the frames there, not everywhere
        out = dist_mod.upscale_sharded(frames.to(runner.dit.device), runner, text, gather="root", **kw)
    else:
        out = pipeline.upscale(frames.to(runner.dit.device), runner, text, **kw)
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.time() - t0
        print(f"Upscaled {out.shape[0]} frames to {out.shape[2]}x{out.shape[1]} in {dt:.2f}s ({out.shape[0] / dt:.2f} FPS, {world} GPU(s))")
    return out, rank


def main(argv: Optional[List[str]] = None) -> int:
    if argv is None and len(sys.argv) == 1:
        sys.argv.append("--help")
    args = build_parser().parse_args(argv)
    devices = [d for d in (getattr(args, "cuda_device", None) or "0").split(",") if d != ""]
    if len(devices) > 1 and "WORLD_SIZE" not in os.environ:
        # multi-GPU: one process per GPU over RCCL (torch.distributed), this script re-run under the launcher
        env = dict(os.environ, HIP_VISIBLE_DEVICES=",".join(devices))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={len(devices)}", "--master-addr",
               "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + (argv if argv is not None else sys.argv[1:])
        return subprocess.call(cmd, env=env)
    if len(devices) == 1 and "WORLD_SIZE" not in os.environ and devices[0] != "0":
        os.environ.setdefault("HIP_VISIBLE_DEVICES", devices[0])
    jobs = list_inputs(args.input)
    for inp in jobs:                                   # the engines stay resident between jobs (interfaces.get_runner)
        frames, fps = load_frames(inp, args.skip_first_frames, args.load_cap)
        out, rank = run(args, frames)
        if rank == 0:
            ext = os.path.splitext(inp)[1].lower()
            fmt = args.output_format or ("mp4" if ext in VIDEO_EXT else "pt" if ext in TENSOR_EXT else "png")
            path = default_output(inp, fmt if not (fmt == "png" and out.shape[0] > 1) else "png_dir")
            if args.output:                            # one job: the path as given; a folder of jobs: a directory to fill
                path = args.output if len(jobs) == 1 else os.path.join(args.output, os.path.basename(path))
            save_frames(out, path, fmt, fps)
            print(f"Saved: {path}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
