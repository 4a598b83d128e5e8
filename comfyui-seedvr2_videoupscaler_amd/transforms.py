"""Phase glue either side of the three runner calls: input transform,
temporal padding, overlap blending.  Device-agnostic tensor plumbing (HBM-bound, no MFMA work);
semantics restated from the reference, results identical on the same inputs:

  video_transform            src/core/generation_utils.py:72-84  (NaResize "side" -> clamp -> DivisiblePad(16)
                              -> Normalize(0.5, 0.5) -> t c h w -> c t h w)
  side_resize / output size   src/data/image/transforms/side_resize.py:37-76 + torchvision
                              ``TVF.resize(img, size:int, BICUBIC, antialias=True)``; torchvision is an
                              un-vendored third-party dependency (requirements.txt): its tensor path is
                              ``torch.nn.functional.interpolate(..., mode="bicubic", align_corners=False,
                              antialias=True)`` on the size ``_compute_resized_output_size`` returns
                              (short edge -> size, long edge -> int(size * long / short))
  divisible_pad               src/data/image/transforms/divisible_crop.py:55-85 (zeros on the right / bottom)
  true_target_dims            src/core/generation_utils.py:124-137 (resize result floored to even)
  pad_video_temporal          src/core/generation_utils.py:598-657 (reversed-frame padding, 4n+1 rule)
  blend_overlapping_frames    src/core/generation_utils.py:284-312 (Hann crossfade for overlap >= 3)
"""
from typing import Tuple

import torch
import torch.nn.functional as F


def resized_output_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision ``_compute_resized_output_size`` for an int ``size`` (shorter edge -> size)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def _bicubic(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    if tuple(x.shape[-2:]) == tuple(size):
        return x
    dt = x.dtype        # torchvision casts every dtype but fp32/fp64 to fp32 around the interpolation (_cast_squeeze_in/out)
    y = F.interpolate(x if dt in (torch.float32, torch.float64) else x.float(),
                      size=list(size), mode="bicubic", align_corners=False, antialias=True)
    return y.to(dt)


def side_resize(x: torch.Tensor, resolution: int, max_resolution: int = 0) -> torch.Tensor:
    """[..., H, W] -> shortest edge = resolution (up- or down-scaling), then no edge above max_resolution."""
    h, w = x.shape[-2:]
    y = _bicubic(x, resized_output_size(h, w, resolution))
    if max_resolution > 0:
        h2, w2 = y.shape[-2:]
        if max(h2, w2) > max_resolution:
            scale = max_resolution / max(h2, w2)
            y = _bicubic(y, (round(h2 * scale), round(w2 * scale)))
    return y


def divisible_pad(x: torch.Tensor, factor: int = 16) -> torch.Tensor:
    h, w = x.shape[-2:]
    ph, pw = (factor - h % factor) % factor, (factor - w % factor) % factor
    if ph == 0 and pw == 0:
        return x
    return F.pad(x, (0, pw, 0, ph), mode="constant", value=0.0)


def true_target_dims(h: int, w: int, resolution: int, max_resolution: int = 0) -> Tuple[int, int]:
    """Output size before the pad-to-16, floored to even (what the decoded frames are trimmed to)."""
    th, tw = resized_output_size(h, w, resolution)
    if max_resolution > 0 and max(th, tw) > max_resolution:
        scale = max_resolution / max(th, tw)
        th, tw = round(th * scale), round(tw * scale)
    return (th // 2) * 2, (tw // 2) * 2


def video_transform(x_tchw: torch.Tensor, resolution: int, max_resolution: int = 0) -> torch.Tensor:
    """[T, C, H, W] in [0, 1] -> [C, T, H', W'] in [-1, 1], H', W' multiples of 16."""
    y = side_resize(x_tchw, resolution, max_resolution)
    y = torch.clamp(y, 0.0, 1.0)
    y = divisible_pad(y, 16)
    y = (y - 0.5) / 0.5
    return y.permute(1, 0, 2, 3)


def pad_video_temporal(videos: torch.Tensor, count: int = 0, temporal_dim: int = 1, prepend: bool = False) -> torch.Tensor:
    """Extend a clip with reversed frames; ``count == 0`` (append) pads to the next 4n+1 length."""
    t = videos.size(temporal_dim)
    if count == 0 and not prepend:
        if t % 4 == 1:
            return videos
        count = ((t - 1) // 4 + 1) * 4 + 1 - t
    if count <= 0:
        return videos

    def sel(a, b):
        return videos.narrow(temporal_dim, a, b - a)

    if count >= t:                      # more padding than frames: repeat the last frame, then mirror
        last = sel(t - 1, t)
        rep = [1] * videos.dim()
        rep[temporal_dim] = count - t + 1
        repeated = last.repeat(*rep)
        mirrored = sel(1, t).flip(temporal_dim) if t > 1 else sel(0, 0)
        parts = [repeated, mirrored, videos] if prepend else [videos, mirrored, repeated]
        return torch.cat(parts, dim=temporal_dim)
    mirrored = sel(1, count + 1).flip(temporal_dim) if prepend else sel(t - count - 1, t - 1).flip(temporal_dim)
    return torch.cat([mirrored, videos] if prepend else [videos, mirrored], dim=temporal_dim)


def blend_overlapping_frames(prev_tail: torch.Tensor, cur_head: torch.Tensor, overlap: int) -> torch.Tensor:
    """[overlap, H, W, C] x 2 -> crossfade (Hann window over the middle third for overlap >= 3, else linear)."""
    dev, dt = prev_tail.device, prev_tail.dtype
    if overlap >= 3:
        t = torch.linspace(0.0, 1.0, steps=overlap, device=dev, dtype=dt)
        u = ((t - 1.0 / 3.0) / (2.0 / 3.0 - 1.0 / 3.0)).clamp(0.0, 1.0)
        w_prev = 0.5 + 0.5 * torch.cos(torch.pi * u)
    else:
        w_prev = torch.linspace(1.0, 0.0, steps=overlap, device=dev, dtype=dt)
    w_prev = w_prev.view(overlap, 1, 1, 1)
    return prev_tail * w_prev + cur_head * (1.0 - w_prev)
