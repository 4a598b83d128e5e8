"""TEST INFRASTRUCTURE (not product code): CPU restatement of the reference's four generation phases for RGB input,
written as one straight-line function over injected components, so that the whole chain -- batching, padding, input
transform, VAE encode, one-step DiT, VAE decode, trims, overlap blend, colour correction, [-1,1] -> [0,1] -- has an
oracle that shares no code with ``<package>/pipeline.py``.

Follows /root/reference/src/core/generation_phases.py:
  encode_all_batches      :300-470   batch boundaries (step = batch_size - overlap, a trailing overlap-only batch is
                                     dropped :349-357), uniform padding :71-105, cast to the compute dtype :377-385,
                                     4n+1 padding :398-404, video_transform :410
  upscale_all_batches     :653-735   seed reset per batch, noise / aug noise, get_condition(task="sr"), runner.inference
                                     (infer.py:315-395: x0 = x_T - v for the one-step schedule, schedules/base.py:108-110)
  decode_all_batches      :935-1010  decode, temporal / spatial trims, overlap blend into the running clip
  postprocess_all_batches :1230-1350 colour correction against the re-transformed input (overlap frames skipped, trimmed
                                     like the sample), clamp, *0.5 + 0.5
Only tests/ and oracle/make_golden.py call this.  ``c`` supplies the components: in make_golden.py they are the reference's
own model classes and the reference's function text (oracle/reference_loader.py); in the CPU tests the in-repo oracles.
"""
from dataclasses import dataclass
from typing import Callable, Optional

import torch


@dataclass
class Components:
    pad_video_temporal: Callable          # (videos, count=0, temporal_dim=1, prepend=False) -> padded
    video_transform: Callable             # [T, C, H, W] in [0, 1] -> [C, T, H', W'] in [-1, 1]
    true_target_dims: Callable            # (h, w) -> (true_h, true_w)
    vae_encode: Callable                  # [C, T, H, W] -> scaled latent [T', h, w, 16]
    dit: Callable                         # (vid [T', h, w, 33], txt) -> v [T', h, w, 16]
    vae_decode: Callable                  # scaled latent -> [C, T, H, W]
    blend_overlapping_frames: Callable    # (prev_tail, cur_head, overlap) -> blended
    color_fix: Optional[Callable]         # (sample [T, C, H, W], reference [T, C, H, W]) -> corrected, or None
    noise: Callable                       # latent -> (base_noise, extra_noise)   (stands in for the seeded randn_like pair)


def upscale(images: torch.Tensor, text: torch.Tensor, c: Components, batch_size: int, temporal_overlap: int = 0,
            uniform_batch_size: bool = False, compute_dtype=torch.float32) -> torch.Tensor:
    """images [T, H, W, 3] in [0, 1] -> [T, H', W', 3] in [0, 1]."""
    total = images.shape[0]
    step = batch_size - temporal_overlap if temporal_overlap > 0 else batch_size
    if step <= 0:
        step, temporal_overlap = batch_size, 0
    true_h, true_w = c.true_target_dims(images.shape[1], images.shape[2])

    # ---- phase 1
    meta, latents, ori_lengths = [], [], []
    for start in range(0, total, step):
        end = min(start + batch_size, total)
        cur = end - start
        if start > 0 and cur <= temporal_overlap:
            break
        pad = batch_size - cur if (uniform_batch_size and cur < batch_size) else 0
        video = images[start:end]
        if pad > 0:
            video = c.pad_video_temporal(video, count=pad, temporal_dim=0, prepend=False)
        video = video.permute(0, 3, 1, 2).to(compute_dtype)
        if video.size(0) % 4 != 1:
            video = c.pad_video_temporal(video.permute(1, 0, 2, 3), temporal_dim=1, prepend=False).permute(1, 0, 2, 3)
        latents.append(c.vae_encode(c.video_transform(video)))
        meta.append((start, end, pad))
        ori_lengths.append(cur)

    # ---- phase 2
    upscaled = []
    for latent in latents:
        base_noise, extra = c.noise(latent)
        _aug = base_noise * 0.1 + extra * 0.05               # (only used with latent_noise_scale > 0; drawn for RNG parity)
        cond = torch.cat([latent, torch.ones_like(latent[..., :1])], dim=-1)        # get_condition, task "sr"
        v = c.dit(torch.cat([base_noise, cond], dim=-1), text)
        upscaled.append(base_noise - v)

    # ---- phase 3
    final, write, spans = None, 0, []
    for i, lat in enumerate(upscaled):
        sample = c.vae_decode(lat)
        if sample.dim() == 3:
            sample = sample.unsqueeze(1)
        sample = sample.permute(1, 0, 2, 3)                  # T C H W
        sample = sample[:ori_lengths[i], :, :true_h, :true_w].permute(0, 2, 3, 1)      # T H W C
        if final is None:
            n_out = sum(l if (j == 0 or temporal_overlap == 0) else max(l - temporal_overlap, 0) for j, l in enumerate(ori_lengths))
            final = torch.zeros(n_out, true_h, true_w, 3, dtype=sample.dtype)
        if i > 0 and temporal_overlap > 0 and temporal_overlap < sample.shape[0] and write >= temporal_overlap:
            final[write - temporal_overlap:write] = c.blend_overlapping_frames(
                final[write - temporal_overlap:write], sample[:temporal_overlap], temporal_overlap)
            sample = sample[temporal_overlap:]
        final[write:write + sample.shape[0]] = sample
        spans.append((write, write + sample.shape[0]))
        write += sample.shape[0]

    # ---- phase 4
    for i, (w0, w1) in enumerate(spans):
        if w1 <= w0:
            continue
        sample = final[w0:w1].permute(0, 3, 1, 2)            # T C H W
        if c.color_fix is not None:
            start, end, pad = meta[i]
            video = images[start:end]
            if pad > 0:
                video = c.pad_video_temporal(video, count=pad, temporal_dim=0, prepend=False)
            video = video.permute(0, 3, 1, 2)
            if video.size(0) % 4 != 1:
                video = c.pad_video_temporal(video.permute(1, 0, 2, 3), temporal_dim=1, prepend=False).permute(1, 0, 2, 3)
            ref = c.video_transform(video).permute(1, 0, 2, 3)
            if i > 0 and temporal_overlap > 0:
                ref = ref[temporal_overlap:]
            ref = ref[:sample.shape[0], :, :true_h, :true_w]
            sample = c.color_fix(sample, ref.to(sample.dtype))
        final[w0:w1] = sample.permute(0, 2, 3, 1).clamp(-1, 1) * 0.5 + 0.5
    return final
