"""ComfyUI node surface of the reference, routed to the MI355X hot path (SURVEY.md 8(b) "API surface to keep").

The reference registers four V3 nodes (src/interfaces/__init__.py:14-29): SeedVR2VideoUpscaler
(video_upscaler.py:52-224), SeedVR2LoadDiTModel (dit_model_loader.py:25-140), SeedVR2LoadVAEModel
(vae_model_loader.py:25-170) and SeedVR2TorchCompileSettings (torch_compile_settings.py:15-100).  Existing workflows
reference them by node id and widget name, so this module keeps ids, widget names, order, defaults, ranges and combo
options (tests/test_api_surface.py compares them with the reference source by ``ast``) and routes ``execute`` to
``checkpoint.build_engines`` + ``pipeline.upscale``.  It is a thin shim, not a port of the reference's model
management: every widget that selects a small-VRAM policy (BlockSwap, offload devices, model caching between runs,
tile debug overlays) or another attention / compiler backend (flash / sage attention, torch.compile) is accepted and
has no effect -- there is one attention kernel and no tracing compiler on this path, and 288 GB of HBM keep both
models and every intermediate tensor resident.

``comfy_api`` is only present inside ComfyUI; without it the schemas are built from plain stand-in records, which is
what the CPU tests inspect.
"""
import os
import threading
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import torch

__version__ = "2.5.24-mi355x"          # the reference release whose surface this mirrors (pyproject.toml) + backend tag

DEFAULT_DIT = "seedvr2_ema_3b_fp8_e4m3fn.safetensors"      # model_registry.py:56-57
DEFAULT_VAE = "ema_vae_fp16.safetensors"
DIT_MODELS = [                                              # model_registry.py:34-52, registry order
    "seedvr2_ema_3b-Q4_K_M.gguf", "seedvr2_ema_3b-Q8_0.gguf", "seedvr2_ema_3b_fp8_e4m3fn.safetensors",
    "seedvr2_ema_3b_fp16.safetensors", "seedvr2_ema_7b-Q4_K_M.gguf",
    "seedvr2_ema_7b_fp8_e4m3fn_mixed_block35_fp16.safetensors", "seedvr2_ema_7b_fp16.safetensors",
    "seedvr2_ema_7b_sharp-Q4_K_M.gguf", "seedvr2_ema_7b_sharp_fp8_e4m3fn_mixed_block35_fp16.safetensors",
    "seedvr2_ema_7b_sharp_fp16.safetensors",
]
VAE_MODELS = ["ema_vae_fp16.safetensors"]
COLOR_CORRECTIONS = ["lab", "wavelet", "wavelet_adaptive", "hsv", "adain", "none"]
ATTENTION_MODES = ["sdpa", "flash_attn_2", "flash_attn_3", "sageattn_2", "sageattn_3"]


# ---------------------------------------------------------------------------------------------------------
@dataclass
class Widget:
    """One node input: kind is the reference's io class ("Image", "Int", "Float", "Boolean", "Combo", "Custom:<TYPE>")."""
    name: str
    kind: str
    default: Any = None
    min: Any = None
    max: Any = None
    step: Any = None
    options: Optional[List[str]] = None
    optional: bool = False


def device_list(include_none: bool = False, include_cpu: bool = False) -> List[str]:
    """memory_manager.get_device_list (src/optimization/memory_manager.py): 'cuda:N' per visible GPU (ROCm devices are
    'cuda' in PyTorch), optionally preceded by 'none' / 'cpu'."""
    devs = [f"cuda:{i}" for i in range(torch.cuda.device_count())] if torch.cuda.is_available() else []
    if not devs:
        devs = ["cpu"]
    head = (["none"] if include_none else []) + (["cpu"] if include_cpu and "cpu" not in devs else [])
    return head + devs


def upscaler_widgets() -> List[Widget]:          # video_upscaler.py:76-217
    return [
        Widget("image", "Image"),
        Widget("dit", "Custom:SEEDVR2_DIT"),
        Widget("vae", "Custom:SEEDVR2_VAE"),
        Widget("seed", "Int", 42, 0, 2 ** 32 - 1, 1),
        Widget("resolution", "Int", 1080, 16, 16384, 2),
        Widget("max_resolution", "Int", 0, 0, 16384, 2),
        Widget("batch_size", "Int", 5, 1, 16384, 4),
        Widget("uniform_batch_size", "Boolean", False),
        Widget("temporal_overlap", "Int", 0, 0, 16, 1, optional=True),
        Widget("prepend_frames", "Int", 0, 0, 32, 1, optional=True),
        Widget("color_correction", "Combo", "lab", options=list(COLOR_CORRECTIONS)),
        Widget("input_noise_scale", "Float", 0.0, 0.0, 1.0, 0.001, optional=True),
        Widget("latent_noise_scale", "Float", 0.0, 0.0, 1.0, 0.001, optional=True),
        Widget("offload_device", "Combo", "cpu", options=device_list(include_none=True, include_cpu=True), optional=True),
        Widget("enable_debug", "Boolean", False, optional=True),
    ]


def dit_loader_widgets() -> List[Widget]:        # dit_model_loader.py:42-128
    devs = device_list()
    return [
        Widget("model", "Combo", DEFAULT_DIT, options=list(DIT_MODELS)),
        Widget("device", "Combo", devs[0], options=devs),
        Widget("blocks_to_swap", "Int", 0, 0, 36, 1, optional=True),
        Widget("swap_io_components", "Boolean", False, optional=True),
        Widget("offload_device", "Combo", "none", options=device_list(include_none=True, include_cpu=True), optional=True),
        Widget("cache_model", "Boolean", False, optional=True),
        Widget("attention_mode", "Combo", "sdpa", options=list(ATTENTION_MODES), optional=True),
        Widget("torch_compile_args", "Custom:TORCH_COMPILE_ARGS", optional=True),
    ]


def vae_loader_widgets() -> List[Widget]:        # vae_model_loader.py:43-157
    devs = device_list()
    return [
        Widget("model", "Combo", DEFAULT_VAE, options=list(VAE_MODELS)),
        Widget("device", "Combo", devs[0], options=devs),
        Widget("encode_tiled", "Boolean", False, optional=True),
        Widget("encode_tile_size", "Int", 1024, 64, None, 32, optional=True),
        Widget("encode_tile_overlap", "Int", 128, 0, None, 32, optional=True),
        Widget("decode_tiled", "Boolean", False, optional=True),
        Widget("decode_tile_size", "Int", 1024, 64, None, 32, optional=True),
        Widget("decode_tile_overlap", "Int", 128, 0, None, 32, optional=True),
        Widget("tile_debug", "Combo", "false", options=["false", "encode", "decode"], optional=True),
        Widget("offload_device", "Combo", "none", options=device_list(include_none=True, include_cpu=True), optional=True),
        Widget("cache_model", "Boolean", False, optional=True),
        Widget("torch_compile_args", "Custom:TORCH_COMPILE_ARGS", optional=True),
    ]


def compile_widgets() -> List[Widget]:           # torch_compile_settings.py:24-88
    return [
        Widget("backend", "Combo", "inductor", options=["inductor", "cudagraphs"]),
        Widget("mode", "Combo", "default", options=["default", "reduce-overhead", "max-autotune", "max-autotune-no-cudagraphs"]),
        Widget("fullgraph", "Boolean", False),
        Widget("dynamic", "Boolean", False),
        Widget("dynamo_cache_size_limit", "Int", 64, 0, 1024, 1),
        Widget("dynamo_recompile_limit", "Int", 128, 0, 1024, 1),
    ]


NODE_TABLE = {
    # node_id: (display name, widget table, output type)
    "SeedVR2VideoUpscaler": (f"SeedVR2 Video Upscaler (v{__version__})", upscaler_widgets, "Image"),
    "SeedVR2LoadDiTModel": ("SeedVR2 (Down)Load DiT Model", dit_loader_widgets, "Custom:SEEDVR2_DIT"),
    "SeedVR2LoadVAEModel": ("SeedVR2 (Down)Load VAE Model", vae_loader_widgets, "Custom:SEEDVR2_VAE"),
    "SeedVR2TorchCompileSettings": ("SeedVR2 Torch Compile Settings", compile_widgets, "Custom:TORCH_COMPILE_ARGS"),
}

# ---------------------------------------------------------------------------------------------------------
try:                                              # inside ComfyUI
    from comfy_api.latest import ComfyExtension, io   # type: ignore
    HAVE_COMFY = True
except Exception:                                 # anywhere else: stand-ins so schemas / execute stay importable and testable
    HAVE_COMFY = False

    class ComfyExtension:                         # noqa: D401
        pass

    class _NodeOutput(tuple):
        def __new__(cls, *values):
            return super().__new__(cls, values)

    @dataclass
    class _Schema:
        node_id: str
        display_name: str
        category: str
        inputs: list
        outputs: list
        description: str = ""

    class _IO:
        ComfyNode = object
        Schema = _Schema
        NodeOutput = _NodeOutput

    io = _IO()


def _to_io_input(w: Widget):
    if not HAVE_COMFY:
        return w
    kw = {k: getattr(w, k) for k in ("default", "min", "max", "step", "options") if getattr(w, k) is not None}
    if w.optional:
        kw["optional"] = True
    if w.kind.startswith("Custom:"):
        return io.Custom(w.kind.split(":", 1)[1]).Input(w.name, **({"optional": True} if w.optional else {}))
    return getattr(io, w.kind).Input(w.name, **kw)


def _to_io_output(kind: str):
    if not HAVE_COMFY:
        return kind
    return io.Custom(kind.split(":", 1)[1]).Output() if kind.startswith("Custom:") else getattr(io, kind).Output()


def _schema(node_id: str):
    display, table, out = NODE_TABLE[node_id]
    return io.Schema(node_id=node_id, display_name=display, category="SEEDVR2",
                     inputs=[_to_io_input(w) for w in table()], outputs=[_to_io_output(out)])


# ---------------------------------------------------------------------------------------------------------
def models_dir() -> str:
    """ComfyUI's models/SEEDVR2 folder (src/utils/constants.py get_base_cache_dir) or $SEEDVR2_MODEL_DIR / ./models/SEEDVR2."""
    if os.environ.get("SEEDVR2_MODEL_DIR"):
        return os.environ["SEEDVR2_MODEL_DIR"]
    try:
        import folder_paths  # type: ignore  (ComfyUI)
        return os.path.join(folder_paths.models_dir, "SEEDVR2")
    except Exception:
        return os.path.join(os.getcwd(), "models", "SEEDVR2")


def resolve_model(name: str, model_dir: Optional[str] = None) -> str:
    path = name if os.path.isabs(name) else os.path.join(model_dir or models_dir(), name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: this backend does not download models (no network on the MI355X hosts); "
                                "place the checkpoint from the reference's model repositories there")
    if path.endswith(".gguf"):
        raise ValueError("GGUF checkpoints are a small-VRAM format and not supported on the MI355X path; use the fp16 / fp8 safetensors")
    return path


def load_text_embedding(device, model_dir: Optional[str] = None) -> torch.Tensor:
    """The reference ships its positive prompt embedding as pos_emb.pt next to the scripts (generation_utils.py:517-557)."""
    here = os.path.dirname(os.path.abspath(__file__))
    for cand in (os.environ.get("SEEDVR2_POS_EMB"), os.path.join(here, "pos_emb.pt"), os.path.join(os.path.dirname(here), "pos_emb.pt"),
                 os.path.join(model_dir or models_dir(), "pos_emb.pt")):
        if cand and os.path.exists(cand):
            return torch.load(cand, map_location="cpu", weights_only=True).to(device=device, dtype=torch.bfloat16)
    raise FileNotFoundError("pos_emb.pt (the reference's prompt embedding, [58, 5120]) not found: copy it from the reference "
                            "checkout next to this package or set SEEDVR2_POS_EMB")


_RUNNERS: Dict[Tuple, Any] = {}                   # (dit path, vae path, device, tiling) -> runner with resident engines
_RUNNERS_LOCK = threading.Lock()                  # ComfyUI may execute nodes from several threads


def get_runner(dit_cfg: Dict[str, Any], vae_cfg: Dict[str, Any], model_dir: Optional[str] = None):
    """Engines resident in HBM, built once per (checkpoints, device) and reused across executions (288 GB make the
    reference's per-run load / offload cycle unnecessary)."""
    from . import checkpoint, ops as ops_mod, runner as runner_mod
    device = dit_cfg.get("device") or "cuda:0"
    dit_path, vae_path = resolve_model(dit_cfg["model"], model_dir), resolve_model(vae_cfg["model"], model_dir)
    tile = (bool(vae_cfg.get("encode_tiled")), int(vae_cfg.get("encode_tile_size", 1024)), int(vae_cfg.get("encode_tile_overlap", 128)),
            bool(vae_cfg.get("decode_tiled")), int(vae_cfg.get("decode_tile_size", 1024)), int(vae_cfg.get("decode_tile_overlap", 128)))
    key = (dit_path, vae_path, device, tile)
    with _RUNNERS_LOCK:
        return _get_runner_locked(key, dit_path, vae_path, device, tile)


def _get_runner_locked(key, dit_path, vae_path, device, tile):
    from . import checkpoint, ops as ops_mod, runner as runner_mod
    if key not in _RUNNERS:
        # one resident model set per process: drop the old one BEFORE building its replacement (two full sets plus the
        # fragment-ordered weight copies would otherwise sit in HBM during the load)
        _RUNNERS.clear()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        ops = ops_mod.HipOps(device)              # raises loudly when the HIP library is missing: no fallback
        dit, vae = checkpoint.build_engines(ops, dit_path, vae_path)
        r = runner_mod.VideoDiffusionInfer(
            runner_mod.default_config(dit.cfg, vae.cfg), encode_tiled=tile[0], encode_tile_size=(tile[1], tile[1]),
            encode_tile_overlap=(tile[2], tile[2]), decode_tiled=tile[3], decode_tile_size=(tile[4], tile[4]),
            decode_tile_overlap=(tile[5], tile[5]))
        r.dit, r.vae = dit, vae
        r.configure_diffusion(device=torch.device(device), dtype=torch.bfloat16)
        _RUNNERS[key] = r
    return _RUNNERS[key]


class SeedVR2TorchCompileSettings(io.ComfyNode):
    @classmethod
    def define_schema(cls):
        return _schema("SeedVR2TorchCompileSettings")

    @classmethod
    def execute(cls, backend: str, mode: str, fullgraph: bool, dynamic: bool, dynamo_cache_size_limit: int,
                dynamo_recompile_limit: int):
        # carried through for workflow compatibility; there is no tracing compiler on the HIP path
        return io.NodeOutput({"backend": backend, "mode": mode, "fullgraph": fullgraph, "dynamic": dynamic,
                              "dynamo_cache_size_limit": dynamo_cache_size_limit,
                              "dynamo_recompile_limit": dynamo_recompile_limit})


class SeedVR2LoadDiTModel(io.ComfyNode):
    @classmethod
    def define_schema(cls):
        return _schema("SeedVR2LoadDiTModel")

    @classmethod
    def execute(cls, model: str, device: str, offload_device: str = "none", cache_model: bool = False,
                blocks_to_swap: int = 0, swap_io_components: bool = False, attention_mode: str = "sdpa",
                torch_compile_args: Optional[Dict[str, Any]] = None):
        if cache_model and offload_device == "none":          # dit_model_loader.py:160-167, same error behaviour
            raise ValueError("Model caching (cache_model=True) requires offload_device to be set. "
                             f"Current: offload_device='{offload_device}'.")
        return io.NodeOutput({"model": model, "device": device, "offload_device": offload_device, "cache_model": cache_model,
                              "blocks_to_swap": blocks_to_swap, "swap_io_components": swap_io_components,
                              "attention_mode": attention_mode, "torch_compile_args": torch_compile_args})


class SeedVR2LoadVAEModel(io.ComfyNode):
    @classmethod
    def define_schema(cls):
        return _schema("SeedVR2LoadVAEModel")

    @classmethod
    def execute(cls, model: str, device: str, offload_device: str = "none", cache_model: bool = False,
                encode_tiled: bool = False, encode_tile_size: int = 512, encode_tile_overlap: int = 64,
                decode_tiled: bool = False, decode_tile_size: int = 512, decode_tile_overlap: int = 64,
                tile_debug: str = "false", torch_compile_args: Optional[Dict[str, Any]] = None):
        # (signature defaults as the reference's execute, vae_model_loader.py:165-171 -- 512 / 64; the WIDGET defaults, which
        #  are what a workflow actually sends, are 1024 / 128)
        if cache_model and offload_device == "none":          # vae_model_loader.py:190-197
            raise ValueError("Model caching (cache_model=True) requires offload_device to be set. "
                             f"Current: offload_device='{offload_device}'.")
        if encode_tiled and encode_tile_overlap >= encode_tile_size:
            raise ValueError(f"VAE encode tile overlap ({encode_tile_overlap}) must be smaller than tile size ({encode_tile_size})")
        if decode_tiled and decode_tile_overlap >= decode_tile_size:
            raise ValueError(f"VAE decode tile overlap ({decode_tile_overlap}) must be smaller than tile size ({decode_tile_size})")
        return io.NodeOutput({"model": model, "device": device, "offload_device": offload_device, "cache_model": cache_model,
                              "encode_tiled": encode_tiled, "encode_tile_size": encode_tile_size,
                              "encode_tile_overlap": encode_tile_overlap, "decode_tiled": decode_tiled,
                              "decode_tile_size": decode_tile_size, "decode_tile_overlap": decode_tile_overlap,
                              "tile_debug": tile_debug, "torch_compile_args": torch_compile_args})


class SeedVR2VideoUpscaler(io.ComfyNode):
    @classmethod
    def define_schema(cls):
        return _schema("SeedVR2VideoUpscaler")

    @classmethod
    def execute(cls, image: torch.Tensor, dit: Dict[str, Any], vae: Dict[str, Any], seed: int, resolution: int = 1080,
                max_resolution: int = 0, batch_size: int = 5, uniform_batch_size: bool = False, temporal_overlap: int = 0,
                prepend_frames: int = 0, color_correction: str = "wavelet", input_noise_scale: float = 0.0,
                latent_noise_scale: float = 0.0, offload_device: str = "none", enable_debug: bool = False):
        """image [N, H, W, C] in [0, 1] -> upscaled [N, H', W', C] in [0, 1] (video_upscaler.py:227-260)."""
        from . import pipeline
        if image.shape[-1] == 4:
            raise NotImplementedError("RGBA input: the alpha path is outside the MI355X hot path (DESIGN.md section 8); pass RGB frames")
        runner = get_runner(dit, vae)
        text = load_text_embedding(runner.dit.device)
        out = pipeline.upscale(image, runner, text, resolution=resolution, max_resolution=max_resolution,
                               batch_size=batch_size, uniform_batch_size=uniform_batch_size,
                               temporal_overlap=temporal_overlap, prepend_frames=prepend_frames,
                               color_correction=color_correction, input_noise_scale=input_noise_scale,
                               latent_noise_scale=latent_noise_scale, seed=seed)
        return io.NodeOutput(out.float().cpu())        # ComfyUI IMAGE tensors live on the host in fp32


class SeedVR2Extension(ComfyExtension):
    async def get_node_list(self):
        return [SeedVR2VideoUpscaler, SeedVR2LoadDiTModel, SeedVR2LoadVAEModel, SeedVR2TorchCompileSettings]


async def comfy_entrypoint():
    return SeedVR2Extension()


__all__ = ["SeedVR2VideoUpscaler", "SeedVR2LoadDiTModel", "SeedVR2LoadVAEModel", "SeedVR2TorchCompileSettings",
           "SeedVR2Extension", "comfy_entrypoint"]
