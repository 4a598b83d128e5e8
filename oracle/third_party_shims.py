"""Minimal stand-ins for the two un-vendored third-party packages the reference
hot path imports, so the reference model code can be imported UNMODIFIED from
/root/reference in this container (test infrastructure only).

The reference pins ``rotary_embedding_torch>=0.5.3`` and ``diffusers>=0.33.1``
(requirements.txt:9,11); neither is installed here and there is no network.
Only the semantics the hot path actually executes are restated:

* rotary_embedding_torch: ``RotaryEmbedding(dim, freqs_for, theta, max_freq)``,
  ``.freqs``, ``.get_axial_freqs(*dims)``, ``apply_rotary_emb(freqs, t)``
  (call sites: src/models/dit_3b/rope.py:19,28-42,76-85,118-126).
* diffusers: ``get_timestep_embedding`` (src/models/dit_3b/embedding.py:17,50-55)
  and the constructor/attribute contracts of the 2D blocks that the 3D VAE
  blocks subclass (src/models/video_vae_v3/modules/attn_video_vae.py:15-28);
  the only third-party ``forward`` that runs is ``Attention.forward``
  (single head, GroupNorm, residual; attn_video_vae.py:615-632,659-665).

The closed forms these restate are unit-tested in tests/test_oracle_shims.py.
"""
import math
import sys
import types
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


# --------------------------------------------------------------------------- #
# rotary_embedding_torch
# --------------------------------------------------------------------------- #
class RotaryEmbedding(nn.Module):
    def __init__(self, dim, freqs_for="lang", theta=10000, max_freq=10, **_):
        super().__init__()
        self.freqs_for = freqs_for
        if freqs_for == "lang":
            freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        elif freqs_for == "pixel":
            freqs = torch.linspace(1.0, max_freq / 2, dim // 2) * math.pi
        else:
            raise ValueError(freqs_for)
        self.freqs = nn.Parameter(freqs, requires_grad=False)

    def forward(self, t):
        f = torch.einsum("..., f -> ... f", t.type(self.freqs.dtype), self.freqs)
        return f.repeat_interleave(2, dim=-1)  # '... n -> ... (n r)', r=2

    def get_axial_freqs(self, *dims):
        all_freqs = []
        for ind, dim in enumerate(dims):
            if self.freqs_for == "pixel":
                pos = torch.linspace(-1, 1, steps=dim, device=self.freqs.device)
            else:
                pos = torch.arange(dim, device=self.freqs.device)
            f = self.forward(pos)
            idx = [None] * len(dims)
            idx[ind] = slice(None)
            all_freqs.append(f[(Ellipsis, *idx, slice(None))])
        all_freqs = torch.broadcast_tensors(*all_freqs)
        return torch.cat(all_freqs, dim=-1)


def rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(-1)
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_rotary_emb(freqs, t, start_index=0, scale=1.0, seq_dim=-2):
    if t.ndim == 3:
        freqs = freqs[-t.shape[seq_dim]:]
    rot = freqs.shape[-1]
    end = start_index + rot
    tl, tm, tr = t[..., :start_index], t[..., start_index:end], t[..., end:]
    tm = (tm * freqs.cos() * scale) + (rotate_half(tm) * freqs.sin() * scale)
    return torch.cat((tl, tm, tr), dim=-1).type(t.dtype)


# --------------------------------------------------------------------------- #
# diffusers
# --------------------------------------------------------------------------- #
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False,
                           downscale_freq_shift=1, scale=1, max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(
        0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :] * scale
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class Attention(nn.Module):
    def __init__(self, query_dim, heads=8, dim_head=64, rescale_output_factor=1.0, eps=1e-5,
                 norm_num_groups=None, spatial_norm_dim=None, residual_connection=False,
                 bias=False, upcast_softmax=False, _from_deprecated_attn_block=False, **_):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.group_norm = (nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True)
                           if norm_num_groups is not None else None)
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, hs, temb=None, **_):
        res = hs
        B, C, H, W = hs.shape
        hs = hs.view(B, C, H * W).transpose(1, 2)
        if self.group_norm is not None:
            hs = self.group_norm(hs.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(hs), self.to_k(hs), self.to_v(hs)
        hd = q.shape[-1] // self.heads
        sp = lambda t: t.view(B, -1, self.heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, -1, self.heads * hd).to(q.dtype)
        o = self.to_out[1](self.to_out[0](o))
        o = o.transpose(-1, -2).reshape(B, C, H, W)
        if self.residual_connection:
            o = o + res
        return o / self.rescale_output_factor


class SpatialNorm(nn.Module):
    pass


class DiagonalGaussianDistribution:
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(
            self.mean.shape, generator=generator, dtype=self.mean.dtype)


@dataclass
class DecoderOutput:
    sample: torch.Tensor
    commit_loss: object = None


@dataclass
class AutoencoderKLOutput:
    latent_dist: object


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv",
                 kernel_size=3, norm_type=None, eps=None, elementwise_affine=None, bias=True):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.padding = padding
        self.name = name
        self.norm = None
        conv = (nn.Conv2d(self.channels, self.out_channels, kernel_size, stride=2,
                          padding=padding, bias=bias) if use_conv else nn.AvgPool2d(2, 2))
        if name == "conv":
            self.Conv2d_0 = conv
        self.conv = conv


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None,
                 name="conv", kernel_size=None, padding=1, norm_type=None, eps=None,
                 elementwise_affine=None, bias=True, interpolate=True):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_conv_transpose = use_conv_transpose
        self.name = name
        self.interpolate = interpolate
        self.norm = None
        conv = (nn.Conv2d(self.channels, self.out_channels, 3, padding=padding, bias=bias)
                if use_conv else None)
        if name == "conv":
            self.conv = conv
        else:
            self.Conv2d_0 = conv


class LoRACompatibleConv(nn.Conv2d):
    pass


class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0,
                 temb_channels=512, groups=32, groups_out=None, pre_norm=True, eps=1e-6,
                 non_linearity="swish", skip_time_act=False, time_embedding_norm="default",
                 kernel=None, output_scale_factor=1.0, use_in_shortcut=None, up=False,
                 down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        self.pre_norm = True
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.up = up
        self.down = down
        self.output_scale_factor = output_scale_factor
        self.time_embedding_norm = time_embedding_norm
        self.skip_time_act = skip_time_act
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.time_emb_proj = (nn.Linear(temb_channels, out_channels)
                              if temb_channels is not None else None)
        self.norm2 = nn.GroupNorm(groups_out, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        conv_2d_out_channels = conv_2d_out_channels or out_channels
        self.conv2 = nn.Conv2d(out_channels, conv_2d_out_channels, 3, 1, 1)
        self.nonlinearity = nn.SiLU()
        self.upsample = self.downsample = None
        if up:
            self.upsample = Upsample2D(in_channels, use_conv=False)
        elif down:
            self.downsample = Downsample2D(in_channels, use_conv=False, padding=1, name="op")
        self.use_in_shortcut = (in_channels != conv_2d_out_channels
                                if use_in_shortcut is None else use_in_shortcut)
        self.conv_shortcut = (nn.Conv2d(in_channels, conv_2d_out_channels, 1, 1, 0,
                                        bias=conv_shortcut_bias)
                              if self.use_in_shortcut else None)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, **_):
        super().__init__()


class UpDecoderBlock2D(nn.Module):
    def __init__(self, **_):
        super().__init__()


class AutoencoderKL(nn.Module):
    def __init__(self, *a, **kw):
        super().__init__()
        self.use_slicing = False
        self.use_tiling = False

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    @property
    def device(self):
        return next(self.parameters()).device


class _RMSNormPlaceholder(nn.Module):
    pass


def install():
    """Register the shim modules in sys.modules (idempotent; never overrides a
    real installation)."""
    if "rotary_embedding_torch" not in sys.modules:
        try:
            import rotary_embedding_torch  # noqa: F401
        except ImportError:
            m = _mod("rotary_embedding_torch")
            m.RotaryEmbedding = RotaryEmbedding
            m.apply_rotary_emb = apply_rotary_emb
            m.rotate_half = rotate_half
    if "diffusers" not in sys.modules:
        try:
            import diffusers  # noqa: F401
            return
        except ImportError:
            pass
        d = _mod("diffusers")
        _mod("diffusers.models")
        ap = _mod("diffusers.models.attention_processor")
        ap.Attention, ap.SpatialNorm = Attention, SpatialNorm
        _mod("diffusers.models.autoencoders")
        v = _mod("diffusers.models.autoencoders.vae")
        v.DecoderOutput, v.DiagonalGaussianDistribution = DecoderOutput, DiagonalGaussianDistribution
        _mod("diffusers.models.downsampling").Downsample2D = Downsample2D
        _mod("diffusers.models.lora").LoRACompatibleConv = LoRACompatibleConv
        _mod("diffusers.models.modeling_outputs").AutoencoderKLOutput = AutoencoderKLOutput
        _mod("diffusers.models.upsampling").Upsample2D = Upsample2D
        _mod("diffusers.models.resnet").ResnetBlock2D = ResnetBlock2D
        _mod("diffusers.models.unets")
        ub = _mod("diffusers.models.unets.unet_2d_blocks")
        ub.DownEncoderBlock2D, ub.UpDecoderBlock2D = DownEncoderBlock2D, UpDecoderBlock2D
        _mod("diffusers.utils").is_torch_version = lambda op, ver: True
        _mod("diffusers.utils.accelerate_utils").apply_forward_hook = lambda f: f
        d.AutoencoderKL = AutoencoderKL
        _mod("diffusers.models.embeddings").get_timestep_embedding = get_timestep_embedding
        _mod("diffusers.models.normalization").RMSNorm = _RMSNormPlaceholder
