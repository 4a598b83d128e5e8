"""Checkpoint ingest: reference checkpoints -> the engines' weight layout.

Replaces ``load_quantized_state_dict`` / ``_load_standard_weights`` (src/core/model_loader.py:84-153, 818-833)
and the dtype handling around them for the formats the hot path supports:

  * ``.safetensors`` (fp16 / bf16 / fp32 / fp8_e4m3fn tensors) and ``.pth``; the key names are the reference
    modules' own (SURVEY.md 8(a) "Weight inventory", 8(c) "Checkpoint key layout"), which is what
    ``NaDiTEngine`` / ``VideoVAEEngine`` consume, so no renaming happens here;
  * FP8 checkpoints: the reference keeps e4m3 weights and lets autocast up-cast them per op
    (compatibility.py:895-938) -- numerically every weight is bf16(e4m3 value), an exact conversion, done
    once here;  fp16 / fp32 weights are cast to bf16 as the reference does at load
    (model_configuration.py:1129-1132);
  * 2D (image-VAE) conv weights are inflated to the causal 3D kernels (``inflate_weight``,
    causal_inflation_lib.py:440-457): "tail" puts the 2D kernel in the last temporal tap, "replicate"
    spreads it / depth;
  * RoPE ``freqs`` buffers missing from a checkpoint are zero-filled like the reference's meta-buffer
    initialisation (model_loader.py:777-815), with a warning;
  * both model families: SeedVR2-3B and -7B (dit_7b) checkpoints are told apart by their tensors.

GGUF (llama.cpp block-quantised) checkpoints are a small-VRAM format and out of scope.
The engines then pre-tile for the MFMA kernels (packing.py).  Registry names: model_registry.py:34-57.
"""
import os
import warnings
from typing import Dict, Iterable, Optional, Tuple

import torch

from .config import DIT_3B, DIT_7B, VAE_V3, DiTConfig, VAEConfig

DEFAULT_DIT = "seedvr2_ema_3b_fp8_e4m3fn.safetensors"      # model_registry.py:56
DEFAULT_VAE = "ema_vae_fp16.safetensors"
BF16 = torch.bfloat16


def load_state_dict(path: str, device: str = "cpu") -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device=str(device))
    if path.endswith(".pth") or path.endswith(".pt"):
        return torch.load(path, map_location=str(device), mmap=True, weights_only=True)
    if path.endswith(".gguf"):
        raise ValueError("GGUF checkpoints are not supported by the MI355X path (use the fp16 / fp8 safetensors)")
    raise ValueError(f"Unsupported checkpoint format. Expected .safetensors or .pth, got: {path}")


def to_compute_dtype(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Every floating tensor -> bf16 (exact for fp8_e4m3fn; what autocast / the load-time cast make of fp16 / fp32
    weights).  RoPE ``freqs`` buffers keep the checkpoint's precision -- the reference only converts them when
    they are stored as fp8 (compatibility.py:787-804) and evaluates the angles in fp32."""
    fp8 = (torch.float8_e4m3fn, torch.float8_e5m2)
    out = {}
    for k, v in sd.items():
        if k.endswith("rope.rope.freqs"):
            out[k] = v.to(BF16) if v.dtype in fp8 else v
        else:
            out[k] = v.to(BF16) if v.is_floating_point() and v.dtype != BF16 else v
    return out


def inflate_weight(weight_2d: torch.Tensor, depth: int, mode: str = "tail") -> torch.Tensor:
    """[Co, Ci, kh, kw] -> [Co, Ci, depth, kh, kw]."""
    if mode not in ("tail", "replicate"):
        raise ValueError("inflation_mode must be 'tail' or 'replicate'")
    co, ci, kh, kw = weight_2d.shape
    if mode == "replicate":
        return weight_2d.unsqueeze(2).repeat(1, 1, depth, 1, 1) / depth
    w3 = torch.zeros(co, ci, depth, kh, kw, dtype=weight_2d.dtype, device=weight_2d.device)
    w3[:, :, -1] = weight_2d
    return w3


def vae_conv_depths(cfg: VAEConfig = VAE_V3) -> Dict[str, int]:
    """Temporal kernel depth of every conv of the causal video VAE (attn_video_vae.py:671-1035): all 3 except the
    first encoder downsampler (1x3x3, spatial only), the 1x1x1 shortcuts and the up-samplers' 1x1x1 upscale convs."""
    n = len(cfg.block_out_channels)
    depths = {}

    def resnet(p):
        depths[p + ".conv1"] = depths[p + ".conv2"] = 3
        depths[p + ".conv_shortcut"] = 1

    for side in ("encoder", "decoder"):
        depths[f"{side}.conv_in"] = depths[f"{side}.conv_out"] = 3
        for j in range(2):
            resnet(f"{side}.mid_block.resnets.{j}")
    for i in range(n):
        for j in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}")
        if i != n - 1:
            depths[f"encoder.down_blocks.{i}.downsamplers.0.conv"] = 3 if i >= n - cfg.temporal_scale_num - 1 else 1
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}")
        if i != n - 1:
            depths[f"decoder.up_blocks.{i}.upsamplers.0.upscale_conv"] = 1
            depths[f"decoder.up_blocks.{i}.upsamplers.0.conv"] = 3
    return depths


def inflate_vae_state_dict(sd: Dict[str, torch.Tensor], cfg: VAEConfig = VAE_V3, mode: str = "tail") -> Dict[str, torch.Tensor]:
    """4-D (2D-conv) weights of known conv layers -> 5-D; already-3D checkpoints pass through unchanged."""
    depths = vae_conv_depths(cfg)
    out = dict(sd)
    for name, d in depths.items():
        w = out.get(name + ".weight")
        if w is not None and w.dim() == 4:
            out[name + ".weight"] = inflate_weight(w, d, mode)
    return out


def dit_expected_keys(cfg: DiTConfig = DIT_3B) -> Iterable[str]:
    from . import weights
    return weights.synth_dit_state_dict(DiTConfig(**{**cfg.as_dict(), "vid_dim": 128, "heads": 1, "txt_in_dim": 64})).keys()


def detect_dit_config(sd: Dict[str, torch.Tensor], name: str = "") -> DiTConfig:
    """SeedVR2-3B or -7B from the tensors themselves (model_registry.py:34-57 keys the choice on the file name; the
    shapes are the safer witness): the 7B family has biased GELU MLPs (``mlp.*.proj_in.bias``) and width 3072."""
    w = sd.get("vid_in.proj.weight", sd.get("model.diffusion_model.vid_in.proj.weight"))
    has_mlp_bias = any(k.endswith("mlp.vid.proj_in.bias") for k in sd)
    if (w is not None and w.shape[0] == DIT_7B.vid_dim) or has_mlp_bias:
        return DIT_7B
    if w is not None and w.shape[0] == DIT_3B.vid_dim:
        return DIT_3B
    if w is None and "7b" in name.lower():
        return DIT_7B
    return DIT_3B


def prepare_dit_state_dict(sd: Dict[str, torch.Tensor], cfg: DiTConfig = DIT_3B) -> Dict[str, torch.Tensor]:
    sd = to_compute_dtype(sd)
    # ComfyUI-style exports may carry a "model.diffusion_model." prefix (model_loader.py:156-160 handles it for GGUF)
    pref = "model.diffusion_model."
    if any(k.startswith(pref) for k in sd):
        sd = {(k[len(pref):] if k.startswith(pref) else k): v for k, v in sd.items()}
    missing = [k for k in dit_expected_keys(cfg) if k not in sd]
    for k in list(missing):
        if k.endswith("rope.rope.freqs"):
            warnings.warn(f"{k} missing from the checkpoint: zero-filled (reference behaviour for meta buffers)")
            sd[k] = torch.zeros(cfg.rope_freqs, dtype=torch.float32)      # 3B: 21 per axis, 7B: 10 (config.DiTConfig.rope_freqs)
            missing.remove(k)
    if missing:
        raise KeyError(f"DiT checkpoint is missing {len(missing)} tensors, e.g. {missing[:5]}")
    return sd


def vae_expected_keys(cfg: VAEConfig = VAE_V3) -> Iterable[str]:
    """Key set of the real architecture (incl. the 1x1x1 ``conv_shortcut`` of every resnet whose width changes), built
    on the meta device: shapes only, no memory."""
    from . import weights
    return weights.synth_vae_state_dict(cfg, device="meta").keys()


def prepare_vae_state_dict(sd: Dict[str, torch.Tensor], cfg: VAEConfig = VAE_V3, inflation_mode: str = "tail") -> Dict[str, torch.Tensor]:
    sd = inflate_vae_state_dict(to_compute_dtype(sd), cfg, inflation_mode)
    missing = [k for k in vae_expected_keys(cfg) if k not in sd]
    if missing:
        raise KeyError(f"VAE checkpoint is missing {len(missing)} tensors, e.g. {missing[:5]}")
    return sd


def build_engines(ops, dit_path: Optional[str] = None, vae_path: Optional[str] = None,
                  dit_cfg: Optional[DiTConfig] = None, vae_cfg: VAEConfig = VAE_V3) -> Tuple[object, object]:
    """(NaDiTEngine | None, VideoVAEEngine | None) from checkpoint files, weights resident in HBM.  ``dit_cfg`` None:
    SeedVR2-3B or -7B is detected from the checkpoint's tensors (detect_dit_config)."""
    from .dit import NaDiTEngine
    from .vae import VideoVAEEngine
    dit = vae = None
    if dit_path:
        sd = load_state_dict(dit_path)
        cfg = dit_cfg or detect_dit_config(sd, os.path.basename(dit_path))
        dit = NaDiTEngine(cfg, prepare_dit_state_dict(sd, cfg), ops)
    if vae_path:
        vae = VideoVAEEngine(vae_cfg, prepare_vae_state_dict(load_state_dict(vae_path), vae_cfg), ops)
    return dit, vae
