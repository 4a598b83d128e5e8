"""Drop-in runner: the object generation_phases.py drives in the reference.

Mirrors ``VideoDiffusionInfer`` (reference: src/core/infer.py:36-395) -- same method names, argument
meaning, ownership and error behaviour -- with ``.dit`` / ``.vae`` backed by the HIP engines:

    runner.vae_encode(samples: List[Tensor[3,T,H,W]])            -> List[Tensor[T',H/8,W/8,16]]   infer.py:117
    runner.vae_decode(latents: List[Tensor[T',h,w,16]])          -> List[Tensor[3,T,H,W]]          infer.py:203
    runner.inference(noises, conditions, texts_pos, texts_neg)   -> List[Tensor[T',h,w,16]]        infer.py:315
    runner.get_condition(latent, latent_blur, task)                                                 infer.py:54
    runner.configure_diffusion(device, dtype); runner.timestep_transform(t, shapes); runner.schedule.forward(...)

The one-step sampler (EulerSampler with steps=1 => a single model call at t = T, x0 = x_t - pred:
euler.py:36-66, schedules/base.py:108-110, lerp.py:44-48, trailing.py:39-48) is folded into the
DiT's un-patchify kernel; several steps and classifier-free guidance (scale, partial, rescale:
diffusion/utils.py:41-86) run as a host loop over the same engine (round 6).  ``config`` is a light attribute tree standing in for the OmegaConf
object (omegaconf is not a dependency here); the keys callers touch are kept.
"""
from typing import List, Optional, Sequence, Tuple, Union

import torch

from .config import DIT_3B, VAE_V3, DiTConfig, VAEConfig

BF16 = torch.bfloat16


class Node(dict):
    """dict with attribute access + .get(), enough of DictConfig for the reference's callers."""
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def default_config(dit: DiTConfig = DIT_3B, vae: VAEConfig = VAE_V3) -> Node:
    """Subset of configs_3b/main.yaml that the pipeline reads or overrides at run time."""
    return Node(
        dit=Node(model=Node(**dit.as_dict())),
        vae=Node(dtype="bfloat16", scaling_factor=vae.scaling_factor, shifting_factor=vae.shifting_factor,
                 grouping=False, use_sample=True,
                 model=Node(spatial_downsample_factor=vae.spatial_downsample_factor,
                            temporal_downsample_factor=vae.temporal_downsample_factor),
                 slicing=Node(split_size=vae.slicing_sample_min_size, memory_device="same")),
        diffusion=Node(schedule=Node(type="lerp", T=1000.0),
                       sampler=Node(type="euler", prediction_type="v_lerp"),
                       timesteps=Node(sampling=Node(type="uniform_trailing", steps=1), transform=True),
                       cfg=Node(scale=1.0, rescale=0.0)),
    )


class LinearInterpolationSchedule:
    """x_t = (1 - t/T) x_0 + (t/T) x_T   (schedules/lerp.py:25-48, base.py:89-96)."""

    def __init__(self, T: float = 1000.0):
        self.T = T

    def A(self, t):
        return 1 - (t / self.T)

    def B(self, t):
        return t / self.T

    def forward(self, x_0: torch.Tensor, x_T: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        while t.dim() < x_0.dim():
            t = t.unsqueeze(-1)
        return self.A(t) * x_0 + self.B(t) * x_T


class VideoDiffusionInfer:
    def __init__(self, config: Optional[Node] = None, debug=None,
                 encode_tiled: bool = False, encode_tile_size: Tuple[int, int] = (512, 512),
                 encode_tile_overlap: Tuple[int, int] = (64, 64),
                 decode_tiled: bool = False, decode_tile_size: Tuple[int, int] = (512, 512),
                 decode_tile_overlap: Tuple[int, int] = (64, 64), tile_debug: str = "false"):
        self.config = config if config is not None else default_config()
        self.debug = debug
        self.encode_tiled, self.encode_tile_size, self.encode_tile_overlap = encode_tiled, encode_tile_size, encode_tile_overlap
        self.decode_tiled, self.decode_tile_size, self.decode_tile_overlap = decode_tiled, decode_tile_size, decode_tile_overlap
        self.tile_debug = tile_debug
        self.dit = None      # NaDiTEngine
        self.vae = None      # VideoVAEEngine
        self.schedule = LinearInterpolationSchedule(self.config.diffusion.schedule.T)

    # ---- infer.py:54-78
    def get_condition(self, latent: torch.Tensor, latent_blur: torch.Tensor, task: str) -> torch.Tensor:
        t, h, w, c = latent.shape
        cond = torch.zeros([t, h, w, c + 1], device=latent.device, dtype=latent.dtype)
        if task == "t2v" or t == 1:
            if task == "sr":
                cond[..., :-1] = latent_blur
                cond[..., -1:] = 1.0
            return cond
        if task == "i2v":
            cond[:1, ..., :-1] = latent[:1]
            cond[:1, ..., -1:] = 1.0
            return cond
        if task == "v2v":
            cond[:2, ..., :-1] = latent[:2]
            cond[:2, ..., -1:] = 1.0
            return cond
        if task == "sr":
            cond[..., :-1] = latent_blur
            cond[..., -1:] = 1.0
            return cond
        raise NotImplementedError

    # ---- infer.py:80-113
    def configure_diffusion(self, device=None, dtype=torch.float32):
        """Schedule + sampling timesteps + sampler as diffusion/config.py:28-75 builds them from ``config.diffusion``: the lerp schedule,
        uniform trailing timesteps (trailing.py:30-50: arange(1, 0, -1/steps), SD3 shift, scaled to T) and the Euler sampler on v_lerp
        predictions -- the only combination the reference's configs use (configs_3b/main.yaml:65-85).  Any number of steps and any
        cfg scale (round 6; the pipeline itself forces steps = 1, cfg = 1.0: generation_phases.py:599-601)."""
        d = self.config.diffusion
        if d.schedule.type != "lerp" or d.sampler.type != "euler" or d.sampler.prediction_type != "v_lerp" \
                or d.timesteps.sampling.type != "uniform_trailing":
            raise NotImplementedError("schedule 'lerp' + sampler 'euler' on 'v_lerp' predictions + 'uniform_trailing' timesteps is the "
                                      "combination the reference's configs use; others are not built")
        steps = int(d.timesteps.sampling.steps)
        if steps < 1:
            raise ValueError("diffusion.timesteps.sampling.steps must be >= 1")
        shift = float(d.timesteps.sampling.get("shift", 1.0))
        T = d.schedule.T
        self.schedule = LinearInterpolationSchedule(T)
        t = torch.arange(1.0, 0.0, -1.0 / steps)
        t = shift * t / (1 + (shift - 1) * t)
        t = t * T if isinstance(T, float) else t.mul(T + 1).sub(1).round().int()
        self.sampling_timesteps = t.to(dtype) if isinstance(T, float) else t

    # ---- infer.py:281-310
    def timestep_transform(self, timesteps: torch.Tensor, latents_shapes: torch.Tensor) -> torch.Tensor:
        if not self.config.diffusion.timesteps.get("transform", False):
            return timesteps
        vt = self.config.vae.model.get("temporal_downsample_factor", 4)
        vs = self.config.vae.model.get("spatial_downsample_factor", 8)
        frames = (latents_shapes[:, 0] - 1) * vt + 1
        heights = latents_shapes[:, 1] * vs
        widths = latents_shapes[:, 2] * vs

        def lin(x1, y1, x2, y2):
            m = (y2 - y1) / (x2 - x1)
            return lambda x: m * x + (y1 - m * x1)

        img_shift = lin(256 * 256, 1.0, 1024 * 1024, 3.2)
        vid_shift = lin(256 * 256 * 37, 1.0, 1280 * 720 * 145, 5.0)
        shift = torch.where(frames > 1, vid_shift(heights * widths * frames), img_shift(heights * widths))
        t = timesteps / self.schedule.T
        t = shift * t / (1 + (shift - 1) * t)
        return t * self.schedule.T

    # ---- infer.py:117-199
    @torch.no_grad()
    def vae_encode(self, samples: List[torch.Tensor]) -> List[torch.Tensor]:
        return [self.vae.encode(s, tiled=self.encode_tiled, tile_size=self.encode_tile_size,
                                tile_overlap=self.encode_tile_overlap) for s in samples]

    # ---- infer.py:203-278
    @torch.no_grad()
    def vae_decode(self, latents: List[torch.Tensor], keep_frames: Optional[Sequence[Optional[int]]] = None) -> List[torch.Tensor]:
        """``keep_frames`` (not in the reference's signature; optional): per latent, the number of leading output frames the
        caller will keep -- the temporal padding it is going to trim is then not decoded (VideoVAEEngine.decode)."""
        keep = list(keep_frames) if keep_frames is not None else [None] * len(latents)
        if len(keep) != len(latents):
            raise ValueError("keep_frames must have one entry per latent")
        return [self.vae.decode(l, tiled=self.decode_tiled, tile_size=self.decode_tile_size,
                                tile_overlap=self.decode_tile_overlap, keep_frames=k) for l, k in zip(latents, keep)]

    # ---- infer.py:315-395
    @torch.no_grad()
    def inference(self, noises: List[torch.Tensor], conditions: List[torch.Tensor],
                  texts_pos: Sequence[torch.Tensor], texts_neg: Sequence[torch.Tensor],
                  cfg_scale: Optional[float] = None) -> List[torch.Tensor]:
        assert len(noises) == len(conditions) == len(texts_pos) == len(texts_neg)
        if len(noises) == 0:
            return []
        if cfg_scale is None:
            cfg_scale = self.config.diffusion.cfg.scale
        cfg_scale = float(cfg_scale)
        if getattr(self, "sampling_timesteps", None) is None:
            self.configure_diffusion()
        ts = [float(t) for t in self.sampling_timesteps.tolist()]
        T = float(self.config.diffusion.schedule.T)
        dev, adt = self.dit.device, self.dit.ops.act_dtype
        outs = []
        if len(ts) == 1 and cfg_scale == 1.0:
            # the pipeline's case: one model call at t = T, the Euler endpoint x_0 = x_T - pred folded into the un-patchify kernel
            for noise, cond, txt in zip(noises, conditions, texts_pos):
                x_t = noise.to(device=dev, dtype=adt).contiguous()
                vid = torch.cat([x_t, cond.to(device=dev, dtype=adt)], dim=-1)          # concat only (infer.py:362)
                outs.append(self.dit.forward(vid, txt.to(device=dev, dtype=adt), timestep=ts[0], x_t=x_t))
            return outs
        # the general sampler (euler.py:36-102, schedules/base.py:82-114, diffusion/utils.py:41-86): every clip of the batch on its own --
        # the reference flattens the batch into one NaDiT call (na.flatten), but clips never interact (windows, text and the timestep
        # embedding are per clip), and classifier-free-guidance rescale takes its standard deviation per token (the flattened [L, C]
        # tensor's dim 1), so the results are the same.  Sampler arithmetic in fp32 on the device; the model sees the activation dtype.
        partial = float(self.config.diffusion.cfg.get("partial", 1))
        rescale = float(self.config.diffusion.cfg.get("rescale", 0.0))
        for noise, cond, tpos, tneg in zip(noises, conditions, texts_pos, texts_neg):
            x = noise.to(device=dev, dtype=torch.float32)
            cond_a = cond.to(device=dev, dtype=adt)
            tpos_a, tneg_a = tpos.to(device=dev, dtype=adt), tneg.to(device=dev, dtype=adt)

            def model(text, t):
                vid = torch.cat([x.to(adt), cond_a], dim=-1)
                return self.dit.forward(vid, text, timestep=t).float()

            for i, t in enumerate(ts):
                scale = cfg_scale if (i + 1) / len(ts) <= partial else 1.0
                pred = model(tpos_a, t)
                if scale != 1.0:                                       # classifier_free_guidance_dispatcher: neg only when needed
                    neg = model(tneg_a, t)
                    cfg = neg + scale * (pred - neg)
                    if rescale != 0.0:                                 # (https://arxiv.org/pdf/2305.08891.pdf) per token, over channels
                        factor = pred.std(dim=-1, keepdim=True) / cfg.std(dim=-1, keepdim=True)
                        cfg = cfg * (rescale * factor + (1 - rescale))
                    pred = cfg
                # v_lerp on the lerp schedule (A + B = 1): x_0 = x_t - B_t pred, x_T = x_t + A_t pred
                x0 = x - (t / T) * pred
                if i + 1 < len(ts):
                    s_ = min(max(ts[i + 1], 0.0), T)
                    xT = x + (1 - t / T) * pred
                    x = (1 - s_ / T) * x0 + (s_ / T) * xT              # step_to: schedule.forward(x_0, x_T, s)
                else:
                    x = x0                                             # return_endpoint: backward direction ends at x_0
            outs.append(x.to(adt))
        return outs
