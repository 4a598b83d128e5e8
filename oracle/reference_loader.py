"""Import the reference's own model code, UNMODIFIED (test infrastructure only; callers must check ``available()`` first):
  * from the checkout at /root/reference where it is mounted (the build container), or
  * from ``oracle/_ref`` -- the same modules byte-compiled by the committed recipe oracle/build_ref.py (no source text; it
    travels to the GPU box with the snapshot), which is how the GPU node times and tests against the reference itself.
``SEEDVR2_REFERENCE_ROOT`` overrides both.

Reference entry points wrapped here:
  NaDiT                         src/models/dit_3b/nadit.py:39
  VideoAutoencoderKLWrapper     src/models/video_vae_v3/modules/attn_video_vae.py:1660
"""
import contextlib
import io
import os
import sys

_COMPILED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _compiled_ok() -> bool:
    """oracle/_ref exists and was compiled by THIS interpreter version (bytecode is version-bound)."""
    import json
    try:
        with open(os.path.join(_COMPILED, "MANIFEST.json")) as f:
            return json.load(f).get("interpreter") == "cpython-%d.%d" % sys.version_info[:2]
    except (OSError, ValueError):
        return False


def _resolve_root() -> str:
    env = os.environ.get("SEEDVR2_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isdir("/root/reference/src/models/dit_3b"):
        return "/root/reference"
    return _COMPILED if _compiled_ok() else "/root/reference"


REFERENCE_ROOT = _resolve_root()


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models", "dit_3b"))


def kind() -> str:
    """"source": the checkout; "compiled": oracle/_ref (the same code, byte-compiled by oracle/build_ref.py)."""
    return "source" if os.path.exists(os.path.join(REFERENCE_ROOT, "src", "models", "dit_3b", "nadit.py")) else "compiled"


def _prepare():
    from . import third_party_shims
    third_party_shims.install()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_nadit_class():
    _prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.dit_3b.nadit import NaDiT
    return NaDiT


def reference_window_module():
    _prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.dit_3b import window
    return window


def reference_vae_class():
    _prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.video_vae_v3.modules.attn_video_vae import VideoAutoencoderKLWrapper
    return VideoAutoencoderKLWrapper


def build_reference_dit_7b(cfg: dict, state_dict=None):
    """The 7B family (src/models/dit_7b/nadit.py:39, configs_7b/main.yaml:11-33) at the given width / depth."""
    import torch
    _prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.dit_7b.nadit import NaDiT
    L = cfg["num_layers"]
    with torch.device("meta") if state_dict is not None else contextlib.nullcontext():
        m = NaDiT(
            vid_in_channels=cfg["vid_in_channels"], vid_out_channels=cfg["vid_out_channels"],
            vid_dim=cfg["vid_dim"], txt_in_dim=cfg["txt_in_dim"], txt_dim=cfg["vid_dim"], emb_dim=6 * cfg["vid_dim"],
            heads=cfg["heads"], head_dim=cfg["head_dim"], expand_ratio=4, norm="fusedrms",
            norm_eps=cfg["norm_eps"], ada="single", qk_bias=False, qk_rope=True, qk_norm="fusedrms",
            patch_size=[1, 2, 2], num_layers=L, shared_mlp=False, shared_qkv=False, mlp_type="normal",
            block_type=["mmdit_sr"] * L, window=[(4, 3, 3)] * L,
            window_method=(["720pwin_by_size_bysize", "720pswin_by_size_bysize"] * L)[:L],
        )
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True, assign=True)
    return m.eval().requires_grad_(False)


def build_reference_dit(cfg: dict, state_dict=None, dtype=None):
    """cfg uses the keys of configs_3b/main.yaml:11-36 (see dit_config() in the package)."""
    import torch
    if cfg.get("rope_type") == "rope3d":
        return build_reference_dit_7b(cfg, state_dict)
    NaDiT = reference_nadit_class()
    L = cfg["num_layers"]
    with torch.device("meta") if state_dict is not None else contextlib.nullcontext():
        m = NaDiT(
            vid_in_channels=cfg["vid_in_channels"], vid_out_channels=cfg["vid_out_channels"],
            vid_dim=cfg["vid_dim"], vid_out_norm="fusedrms", txt_in_dim=cfg["txt_in_dim"],
            txt_in_norm="fusedln", txt_dim=cfg["vid_dim"], emb_dim=6 * cfg["vid_dim"],
            heads=cfg["heads"], head_dim=cfg["head_dim"], expand_ratio=4, norm="fusedrms",
            norm_eps=cfg["norm_eps"], ada="single", qk_bias=False, qk_norm="fusedrms",
            patch_size=[1, 2, 2], num_layers=L, mm_layers=cfg["mm_layers"], mlp_type="swiglu",
            msa_type=None, block_type=["mmdit_sr"] * L, window=[(4, 3, 3)] * L,
            window_method=(["720pwin_by_size_bysize", "720pswin_by_size_bysize"] * L)[:L],
            rope_type="mmrope3d", rope_dim=128,
        )
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True, assign=True)
    m = m.eval().requires_grad_(False)
    if dtype is not None:
        m = m.to(dtype)
    return m


def reference_sampler_stack(T=1000.0, steps=1, shift=1.0, prediction_type="v_lerp"):
    """The reference's diffusion stack as VideoDiffusionInfer.configure_diffusion builds it (infer.py:80-113, diffusion/config.py:28-75):
    LinearInterpolationSchedule, UniformTrailingSamplingTimesteps, EulerSampler -- plus the classifier-free-guidance dispatcher
    (diffusion/utils.py:41-86) and the na.flatten / na.unflatten pair inference() batches with (infer.py:355-386).
    ``src/common/diffusion/config.py`` imports omegaconf for a type annotation only; where omegaconf is absent a module holding that
    one name is registered so that the package imports (none of the factory functions that would read a DictConfig is called)."""
    import types
    _prepare()
    try:
        import omegaconf  # noqa: F401
    except ImportError:
        m = types.ModuleType("omegaconf")
        m.DictConfig, m.ListConfig, m.OmegaConf = dict, list, object
        sys.modules["omegaconf"] = m
    with contextlib.redirect_stdout(io.StringIO()):
        from src.common.diffusion.samplers.euler import EulerSampler
        from src.common.diffusion.schedules.lerp import LinearInterpolationSchedule
        from src.common.diffusion.timesteps.sampling.trailing import UniformTrailingSamplingTimesteps
        from src.common.diffusion.utils import classifier_free_guidance_dispatcher
        from src.models.dit_3b import na
    schedule = LinearInterpolationSchedule(T=T)
    timesteps = UniformTrailingSamplingTimesteps(T=T, steps=steps, shift=shift)
    sampler = EulerSampler(schedule=schedule, timesteps=timesteps, prediction_type=prediction_type)
    sampler.get_progress_bar = lambda: type("P", (), {"update": lambda self: None})()      # (tqdm console output only)
    return {"schedule": schedule, "timesteps": timesteps, "sampler": sampler, "cfg": classifier_free_guidance_dispatcher, "na": na}


VAE_CFG = dict(  # src/models/video_vae_v3/s8_c16_t4_inflation_sd3.yaml
    act_fn="silu", block_out_channels=[128, 256, 512, 512],
    down_block_types=["DownEncoderBlock3D"] * 4, in_channels=3, latent_channels=16,
    layers_per_block=2, norm_num_groups=32, out_channels=3, slicing_sample_min_size=4,
    temporal_scale_num=2, inflation_mode="pad", up_block_types=["UpDecoderBlock3D"] * 4,
    spatial_downsample_factor=8, temporal_downsample_factor=4, use_quant_conv=False,
    use_post_quant_conv=False, freeze_encoder=False, gradient_checkpoint=True,
)


def build_reference_vae(state_dict=None, dtype=None, block_out_channels=None, slicing=True):
    import torch
    cls = reference_vae_class()
    cfg = dict(VAE_CFG)
    if block_out_channels is not None:
        cfg["block_out_channels"] = list(block_out_channels)
    with torch.device("meta") if state_dict is not None else contextlib.nullcontext():
        vae = cls(**cfg)
    if state_dict is not None:
        vae.load_state_dict(state_dict, strict=True, assign=True)
    vae = vae.eval().requires_grad_(False)
    if dtype is not None:
        vae = vae.to(dtype)
    vae.debug = None                      # normally set at model_configuration.py:1273-1274
    vae.tensor_offload_device = None
    if slicing:                            # configs_3b/main.yaml:53-58
        vae.set_causal_slicing(split_size=4, memory_device="same")
        vae.set_memory_limit(conv_max_mem=0.5, norm_max_mem=0.5)
    return vae


# ---------------------------------------------------------------------------------------------------------
# Phase-glue functions.  Their modules import torchvision / omegaconf / cv2, which are
# not installed here, so the *function definitions* are compiled straight from the reference source files
# (unmodified text, selected by name with ``ast``) into a namespace that only provides what they touch.
def _extract(path: str, names, namespace: dict) -> dict:
    import ast
    if not os.path.exists(os.path.join(REFERENCE_ROOT, path)):
        # oracle/_ref: the definitions' code objects, compiled from the unmodified AST by oracle/build_ref.py
        import marshal
        with open(os.path.join(REFERENCE_ROOT, "defs", path.replace("/", "__") + ".marshal"), "rb") as f:
            table = marshal.load(f)
        missing = set(names) - set(table)
        if missing:
            raise ImportError(f"{path}: {sorted(missing)} not found")
        for name, code in table.items():               # (file order, as the source branch below)
            if name in names:
                exec(code, namespace)
        return namespace
    src = open(os.path.join(REFERENCE_ROOT, path)).read()
    tree = ast.parse(src)
    picked = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise ImportError(f"{path}: {sorted(missing)} not found")
    mod = ast.Module(body=picked, type_ignores=[])
    exec(compile(mod, os.path.join(REFERENCE_ROOT, path), "exec"), namespace)
    return namespace


def reference_glue():
    """dict with the reference's pad_video_temporal, blend_overlapping_frames, SideResize, DivisiblePad and the
    colour-fix functions.  ``TVF.resize`` is bound to the restatement of torchvision's tensor path (third-party,
    un-vendored: see transforms.py header) -- everything else runs from the reference text."""
    import typing
    import torch
    import torch.nn.functional as F
    from PIL import Image

    class _Interp:          # torchvision.transforms.InterpolationMode stand-in
        BICUBIC, BILINEAR = "bicubic", "bilinear"

    class _TVF:
        @staticmethod
        def resize(img, size, interpolation=_Interp.BICUBIC, antialias=True):
            h, w = img.shape[-2:]
            if isinstance(size, int):
                short, long = (w, h) if w <= h else (h, w)
                new_short, new_long = size, int(size * long / short)
                size = (new_long, new_short) if w <= h else (new_short, new_long)
            dt = img.dtype
            x = img if dt in (torch.float32, torch.float64) else img.float()
            squeeze = x.dim() == 3
            if squeeze:
                x = x.unsqueeze(0)
            y = F.interpolate(x, size=list(size), mode=interpolation, align_corners=False, antialias=antialias)
            return (y.squeeze(0) if squeeze else y).to(dt)

    ns = {"torch": torch, "F": F, "Tensor": torch.Tensor, "Image": Image, "Union": typing.Union,
          "Optional": typing.Optional, "Dict": typing.Dict, "Any": typing.Any, "InterpolationMode": _Interp,
          "TVF": _TVF, "is_mps_available": lambda: False,
          "safe_pad_operation": lambda x, pad, mode="constant", value=0.0: F.pad(x, pad, mode=mode, value=value)
          if mode == "constant" else F.pad(x, pad, mode=mode),
          "safe_interpolate_operation": lambda x, **kw: F.interpolate(x, **kw),
          "ensure_float32_precision": lambda t: (t.float(), t.dtype)}
    _extract("src/core/generation_utils.py", ["pad_video_temporal", "blend_overlapping_frames"], ns)
    _extract("src/data/image/transforms/side_resize.py", ["SideResize"], ns)
    _extract("src/data/image/transforms/divisible_crop.py", ["DivisiblePad"], ns)
    _extract("src/utils/color_fix.py",
             ["calc_mean_std", "adaptive_instance_normalization", "wavelet_blur", "wavelet_decomposition",
              "wavelet_reconstruction", "lab_color_transfer", "_rgb_to_lab_batch", "_lab_to_rgb_batch",
              "_histogram_matching_channel", "hsv_saturation_histogram_match", "_rgb_to_hsv_batch", "_hsv_to_rgb_batch",
              "_hue_conditional_saturation_match", "_histogram_match_1d", "wavelet_adaptive_color_correction",
              "_get_saturation_map"], ns)
    return ns


# ---------------------------------------------------------------------------------------------------------
# The reference's four phase functions THEMSELVES (generation_phases.py:171 encode_all_batches, :542 upscale_all_batches,
# :807 decode_all_batches, :1060 postprocess_all_batches + their three private helpers), compiled from the unmodified source
# text, so a test can drive them over this repo's runner (the drop-in claim of INTEGRATION.md).  What the namespace provides
# instead of the reference's own modules, and why:
#   * torchvision (third-party, not installed): Compose / Lambda / Normalize stand-ins, TVF.resize as in reference_glue();
#   * memory management (manage_tensor, manage_model_device, release_*, cleanup_*): host-RAM / VRAM policy, out of scope
#     (SURVEY.md 8 "out of scope") -- manage_tensor keeps its contract (move + cast), the others are no-ops;
#   * materialize_model / apply_model_specific_config / load_text_embeddings / process_alpha_for_batch: model management and
#     the alpha path -- they raise if the phases ever reach them;
#   * everything else (pad_video_temporal, blend_overlapping_frames, setup_video_transform, prepare_video_transforms,
#     calculate_optimal_batch_params, check_interrupt, ensure_precision_initialized, set_seed, the optimized_* rearranges,
#     NaResize / SideResize / DivisiblePad, the colour-fix functions) runs from the reference text.
class PhaseDebug:
    """Debug stand-in: swallows the phases' logging / timing calls (src/utils/debug.py is console output only)."""
    encode_tile_boundaries = None
    decode_tile_boundaries = None

    def __getattr__(self, name):
        return lambda *a, **k: None


def reference_phases(torch_module=None):
    """-> namespace dict with the four phase functions.  ``torch_module``: what the phases see as ``torch`` (a proxy lets a
    test inject the noise a golden was made with); default: torch itself."""
    import typing
    import random
    import numpy as np
    import torch
    ns = reference_glue()
    tm = torch_module if torch_module is not None else torch

    class Compose:
        def __init__(self, transforms):
            self.transforms = list(transforms)

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class Lambda:
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, x):
            return self.fn(x)

    class Normalize:                                   # torchvision.transforms.Normalize on a [..., C, H, W] float tensor
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            return (x - self.mean) / self.std

    def manage_tensor(tensor, target_device, tensor_name="", dtype=None, non_blocking=False, debug=None, reason=None,
                      indent_level=0):
        return tensor.to(device=target_device, dtype=dtype if dtype is not None else tensor.dtype)

    def _never(name):
        def f(*a, **k):
            raise AssertionError(f"{name}() is model management / alpha handling: the phases must not need it over a ready runner")
        return f

    noop = lambda *a, **k: None
    ns.update({"torch": tm, "os": os, "Dict": typing.Dict, "List": typing.List, "Optional": typing.Optional, "Tuple": typing.Tuple,
               "Any": typing.Any, "Callable": typing.Callable, "Union": typing.Union, "Literal": typing.Literal,
               "random": random, "np": np, "get_global_rank": lambda: 0,
               "Compose": Compose, "Lambda": Lambda, "Normalize": Normalize, "AreaResize": None, "Resize": None, "CenterCrop": None,
               "manage_tensor": manage_tensor, "manage_model_device": noop, "release_tensor_memory": noop,
               "release_tensor_collection": noop, "cleanup_dit": noop, "cleanup_vae": noop, "cleanup_text_embeddings": noop,
               "materialize_model": _never("materialize_model"), "apply_model_specific_config": _never("apply_model_specific_config"),
               "load_text_embeddings": _never("load_text_embeddings"), "process_alpha_for_batch": _never("process_alpha_for_batch"),
               "_draw_tile_boundaries": _never("_draw_tile_boundaries"), "script_directory": None})
    _extract("src/common/seed.py", ["set_seed"], ns)
    _extract("src/optimization/performance.py",
             ["optimized_video_rearrange", "optimized_single_video_rearrange", "optimized_sample_to_image_format"], ns)
    _extract("src/data/image/transforms/na_resize.py", ["NaResize"], ns)
    _extract("src/core/generation_utils.py",
             ["prepare_video_transforms", "setup_video_transform", "calculate_optimal_batch_params", "check_interrupt",
              "ensure_precision_initialized"], ns)
    _extract("src/core/generation_phases.py",
             ["_prepare_video_batch", "_apply_4n1_padding", "_reconstruct_and_transform_batch", "encode_all_batches",
              "upscale_all_batches", "decode_all_batches", "postprocess_all_batches"], ns)
    return ns


def phase_context(device, compute_dtype, text):
    """The keys of setup_generation_context's dict (generation_utils.py:315-419) that the four phases read, for a runner whose
    models are resident: no offload devices, no model cache, no interrupt hook."""
    import torch
    dev = torch.device(device)
    return {"vae_device": dev, "dit_device": dev, "vae_offload_device": None, "dit_offload_device": None,
            "tensor_offload_device": None, "compute_dtype": compute_dtype, "interrupt_fn": None, "video_transform": None,
            "text_embeds": {"texts_pos": [text], "texts_neg": [text]},
            "cache_context": {"vae_cache": False, "dit_cache": False, "cached_vae": None, "cached_dit": None,
                              "vae_newly_cached": False, "dit_newly_cached": False}}
