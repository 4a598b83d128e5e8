"""The four generation phases around the runner, kept entirely on the device.

Mirrors ``encode_all_batches -> upscale_all_batches -> decode_all_batches -> postprocess_all_batches``
(reference: src/core/generation_phases.py:171-540, 542-805, 807-1058, 1060-1400) for RGB input:

  * batching with ``temporal_overlap`` (step = batch_size - overlap; a trailing batch that holds only overlap
    frames is dropped, :349-357), ``uniform_batch_size`` padding with reversed frames (:71-105), the 4n+1
    rule (:398-404), the input transform (generation_utils.py:72-84) and optional input noise (:413-428);
  * per batch: seed reset (:663), ``noise = randn_like(latent)``, ``aug_noise = noise*0.1 + randn*0.05``,
    optional latent noise through the schedule (:680-697), ``get_condition(task="sr")``, one DiT step;
  * decode, trim to the original batch length and to ``true_target_dims`` (:953-968), Hann blend of the
    overlapping frames (:973-1000), write into the preallocated output;
  * colour correction against the re-transformed input (:1255-1320), ``[-1, 1] -> [0, 1]`` (:1348).

Differences by design (MI355X-first, results unchanged): nothing leaves the GPU between phases (the reference
round-trips every latent through host RAM by default) and with 288 GB the default is to keep the
whole output clip in HBM.  Temporal batches are independent units, which is what ``dist.py`` shards over ranks.
"""
from dataclasses import dataclass
import inspect
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import colorfix, transforms


@dataclass
class BatchPlan:
    start: int          # first input frame of the batch
    end: int            # one past the last input frame
    uniform_pad: int    # reversed frames appended so that every batch has batch_size frames


def plan_batches(total_frames: int, batch_size: int, temporal_overlap: int = 0,
                 uniform_batch_size: bool = False) -> Tuple[List[BatchPlan], int]:
    """Batch boundaries of encode_all_batches (generation_phases.py:300-360).  Returns (plans, overlap actually used)."""
    if total_frames <= 0:
        raise ValueError("No frames to process")
    step = batch_size - temporal_overlap if temporal_overlap > 0 else batch_size
    if step <= 0:
        step, temporal_overlap = batch_size, 0
    plans = []
    for idx in range(0, total_frames, step):
        end = min(idx + batch_size, total_frames)
        if idx > 0 and end - idx <= temporal_overlap:
            break
        cur = end - idx
        plans.append(BatchPlan(idx, end, batch_size - cur if (uniform_batch_size and cur < batch_size) else 0))
    return plans, temporal_overlap


def prepare_batch(images_thwc: torch.Tensor, plan: BatchPlan, resolution: int, max_resolution: int = 0,
                  dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """frames [start, end) -> uniform padding -> 4n+1 padding -> input transform: [3, T', H', W'] in [-1, 1].
    ``dtype``: cast the frames BEFORE padding / resizing, as phase 1 of the reference does (the batch goes to the
    compute dtype at generation_phases.py:377-385, so its resize result, clamp, pad and normalise are rounded to bf16
    step by step); phase 4 re-transforms the input in its own dtype (generation_phases.py:127-168) -> ``None``."""
    video = images_thwc[plan.start:plan.end]
    if dtype is not None:
        video = video.to(dtype)
    if plan.uniform_pad > 0:
        video = transforms.pad_video_temporal(video, count=plan.uniform_pad, temporal_dim=0, prepend=False)
    video = video.permute(0, 3, 1, 2)                                    # T C H W
    if video.size(0) % 4 != 1:
        video = transforms.pad_video_temporal(video, temporal_dim=0, prepend=False)
    return transforms.video_transform(video[:, :3], resolution, max_resolution)


def transformed_shape(images_thwc: torch.Tensor, plan: BatchPlan, resolution: int, max_resolution: int = 0) -> Tuple[int, ...]:
    """Shape of prepare_batch()'s result without computing it: [3, T', H', W']."""
    t = plan.end - plan.start + plan.uniform_pad
    if t % 4 != 1:
        t = ((t - 1) // 4 + 1) * 4 + 1
    h, w = transforms.resized_output_size(images_thwc.shape[1], images_thwc.shape[2], resolution)
    if max_resolution > 0 and max(h, w) > max_resolution:
        scale = max_resolution / max(h, w)
        h, w = round(h * scale), round(w * scale)
    return (3, t, (h + 15) // 16 * 16, (w + 15) // 16 * 16)


def _takes_keep_frames(runner) -> bool:
    """Only this repo's runner knows ``vae_decode(..., keep_frames=)``; any object with the reference's three calls still works."""
    try:
        return "keep_frames" in inspect.signature(runner.vae_decode).parameters
    except (TypeError, ValueError):
        return False


@torch.no_grad()
def upscale(images_thwc: torch.Tensor, runner, text_pos: torch.Tensor, *, resolution: int = 1080,
            max_resolution: int = 0, batch_size: int = 5, uniform_batch_size: bool = False,
            temporal_overlap: int = 0, prepend_frames: int = 0, color_correction: str = "lab",
            input_noise_scale: float = 0.0, latent_noise_scale: float = 0.0, seed: int = 42,
            batch_filter: Optional[Callable[[int], bool]] = None,
            progress: Optional[Callable[[str, int, int], None]] = None,
            noise_provider: Optional[Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]] = None,
            exchange_heads: Optional[Callable[[dict, list, tuple, torch.dtype], dict]] = None,
            return_spans: bool = False, skip_trimmed_frames: bool = True, output_dtype: Optional[torch.dtype] = torch.float32):
    """images [T, H, W, 3] in [0, 1] (any float dtype, on the runner's device) -> upscaled [T, H', W', 3] in [0, 1].

    ``batch_filter(i)`` restricts phases 1-3 to the temporal batches a rank owns (data parallelism over
    batches); frames of skipped batches are left at zero for the caller's all-gather/sum.
    ``output_dtype`` (default fp32 = ComfyUI's IMAGE dtype; +0.7 dB at production width): what the decoded frames
    are held in from the decoder's output on -- the clip buffer ``final``, the overlap heads exchanged between ranks and the
    gathered clip of dist.upscale_sharded.  Memory / xGMI cost: 99.5 MB per 4K frame in fp32 against 49.8 MB in bf16 (a 128-frame
    4K clip: 12.7 GB per rank, 1.6 GB per rank on the wire at 8 ranks -- sized for 288 GB of HBM); ``output_dtype=None`` keeps the
    engines' activation dtype (bf16) as the reference's phase code does, for clips where that matters more than the 0.7 dB.
    """
    if images_thwc.shape[-1] != 3:
        raise NotImplementedError("RGB input only (the alpha path is outside the hot path, DESIGN.md section 7)")
    # compute / storage dtype of the phases = the engines' activation dtype (bf16 on the HIP path, as the reference's
    # compute_dtype; fp32 when the CPU tests drive the engines with the fp32 torch double of the C ABI)
    dev, dt = runner.dit.device, getattr(getattr(runner.dit, "ops", None), "act_dtype", torch.bfloat16)
    # what the decoded frames are held in from the decoder's output on (trims, overlap blend, colour fix, [-1, 1] -> [0, 1]) and
    # returned in.  Default fp32 = ComfyUI's IMAGE dtype (video_upscaler.py:241): a bf16 [0, 1] frame has a step of 2^-8 above 0.5,
    # i.e. 1.1e-3 rms of rounding per stage -- 1.3e-6 of the 1e-5 MSE that 50 dB allows (tools/error_budget.py);
    # None: the engines' activation dtype (rounds 2-3; the reference keeps final_video in its compute dtype)
    odt = output_dtype if output_dtype is not None else dt
    images = images_thwc.to(device=dev)
    if prepend_frames > 0:
        images = transforms.pad_video_temporal(images, count=prepend_frames, temporal_dim=0, prepend=True)
    total = images.shape[0]
    true_h, true_w = transforms.true_target_dims(images.shape[1], images.shape[2], resolution, max_resolution)
    plans, overlap = plan_batches(total, batch_size, temporal_overlap, uniform_batch_size)
    mine = [i for i in range(len(plans)) if batch_filter is None or batch_filter(i)]

    # ---- phase 1: encode.  The input noise comes from ONE generator stream seeded with seed + 1e6 before the first
    # batch (set_seed(seed_vae), generation_phases.py:327-330) and consumed batch after batch; a rank that skips batches
    # (batch_filter) draws and discards their noise, so every batch sees the noise of the single-rank run.
    latents = {}
    if input_noise_scale > 0:
        torch.manual_seed(seed + 1_000_000)
        if dev.type == "cuda":
            torch.cuda.manual_seed(seed + 1_000_000)
    owned = set(mine)
    done = 0
    for i, plan in enumerate(plans):
        if i not in owned:
            if input_noise_scale > 0 and i < max(mine, default=-1):
                c, t, h, w = transformed_shape(images, plan, resolution, max_resolution)
                # same shape AND strides as prepare_batch's result (a [T, C, H, W] buffer viewed c t h w): the CPU generator
                # consumes its stream differently for contiguous and strided outputs
                torch.randn_like(torch.empty((t, c, h, w), dtype=dt, device=dev).permute(1, 0, 2, 3))
            continue
        x = prepare_batch(images, plan, resolution, max_resolution, dtype=dt)
        if input_noise_scale > 0:
            noise = torch.randn_like(x) * 0.05
            blend = input_noise_scale * 0.5
            x = x * (1 - blend) + (x + noise) * blend
        latents[i] = runner.vae_encode([x])[0]
        done += 1
        if progress:
            progress("encode", done, len(mine))

    # ---- phase 2: one-step DiT
    upscaled = {}
    for n, i in enumerate(mine):
        latent = latents.pop(i).to(dt)
        torch.manual_seed(seed)                        # identical RNG state for every batch (generation_phases.py:663)
        if dev.type == "cuda":
            torch.cuda.manual_seed(seed)
        if noise_provider is None:
            base_noise = torch.randn_like(latent)
            extra = torch.randn_like(base_noise)
        else:
            base_noise, extra = (t.to(device=dev, dtype=dt) for t in noise_provider(latent))
        aug_noise = base_noise * 0.1 + extra * 0.05
        blur = latent
        if latent_noise_scale != 0.0:
            t = torch.tensor([1000.0], device=dev, dtype=dt) * latent_noise_scale
            shape = torch.tensor(latent.shape[1:], device=dev)[None]
            blur = runner.schedule.forward(latent, aug_noise, runner.timestep_transform(t, shape))
        cond = runner.get_condition(base_noise, task="sr", latent_blur=blur)
        upscaled[i] = runner.inference([base_noise], [cond], [text_pos], [text_pos])[0]
        if progress:
            progress("upscale", n + 1, len(mine))

    # ---- phase 3: decode, trim, blend into the output clip ([-1, 1] until phase 4)
    final = torch.zeros(total, true_h, true_w, 3, dtype=odt, device=dev)
    spans, heads, starts = {}, {}, {}
    write = 0
    for i, plan in enumerate(plans):
        starts[i] = write
        ori = plan.end - plan.start
        n_new = ori if (i == 0 or overlap == 0) else max(ori - overlap, 0)
        if i in upscaled:
            # (skip_trimmed_frames: the decoder is causal in time, so our runner leaves out the padding frames that are trimmed
            # two lines down -- same result, bit for bit on the HIP path (tests/test_gpu_parity.py::test_vae_decode_keep_frames_bit_exact))
            sample = (runner.vae_decode([upscaled.pop(i)], keep_frames=[ori])[0]
                      if skip_trimmed_frames and _takes_keep_frames(runner) else runner.vae_decode([upscaled.pop(i)])[0])
            if sample.dim() == 3:
                sample = sample.unsqueeze(1)
            sample = sample.permute(1, 2, 3, 0)[:ori, :true_h, :true_w].to(odt)           # T H W C, padding trimmed
            if i > 0 and 0 < overlap < sample.shape[0] and write >= overlap:
                if (i - 1) in spans:                                                      # both sides are on this rank
                    final[write - overlap:write] = transforms.blend_overlapping_frames(
                        final[write - overlap:write], sample[:overlap], overlap)
                else:
                    heads[i] = sample[:overlap].contiguous()                              # blended by the previous batch's owner
                sample = sample[overlap:]
            final[write:write + sample.shape[0]] = sample
            spans[i] = (write, write + sample.shape[0])
            if progress:
                progress("decode", len(spans), len(mine))
        write += n_new

    if exchange_heads is not None and overlap > 0:
        # batch boundaries that carry an overlap blend -- a function of the plans alone, so every rank derives the same list
        # and the point-to-point exchange posts matching sends and receives
        boundaries = [i for i, plan in enumerate(plans) if i > 0 and 0 < overlap < plan.end - plan.start and starts[i] >= overlap]
        for i, head in exchange_heads(heads, boundaries, (overlap, true_h, true_w, 3), odt).items():    # (heads travel in the frames' dtype)
            if (i - 1) in spans and i not in spans:
                w = starts[i]
                final[w - overlap:w] = transforms.blend_overlapping_frames(final[w - overlap:w], head.to(final), overlap)

    # ---- phase 4: colour correction against the re-transformed input, [-1, 1] -> [0, 1]
    for i, (w0, w1) in spans.items():
        if w1 <= w0:
            continue
        sample = final[w0:w1].permute(0, 3, 1, 2)
        if color_correction != "none":
            ref = prepare_batch(images, plans[i], resolution, max_resolution).to(odt).permute(1, 0, 2, 3)   # T C H W
            if i > 0 and overlap > 0:
                ref = ref[overlap:]
            ref = ref[:sample.shape[0], :, :true_h, :true_w]
            if color_correction not in colorfix.METHODS:
                raise ValueError(f"Unknown color correction method: {color_correction}")
            sample = colorfix.METHODS[color_correction](sample, ref)
        final[w0:w1] = sample.permute(0, 2, 3, 1).clamp(-1, 1).mul(0.5).add(0.5).to(odt)
    if prepend_frames > 0:
        final = final[prepend_frames:]
        spans = {i: (max(a - prepend_frames, 0), max(b - prepend_frames, 0)) for i, (a, b) in spans.items()}
    return (final, spans) if return_spans else final
