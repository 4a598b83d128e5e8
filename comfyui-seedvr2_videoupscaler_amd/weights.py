"""Synthetic, seeded SeedVR2 weights under the reference's own state-dict key names.

No SeedVR2 checkpoint can be fetched in this environment, so parity and throughput are
measured on random-initialised weights of the exact architecture; the key names / shapes
below are the ones the reference modules register (probed on the meta device, see
SURVEY.md section 8(a) "Weight inventory" and 8(c) "Checkpoint key layout"), so a real
``seedvr2_ema_3b_*.safetensors`` / ``ema_vae_*.safetensors`` state dict can be passed to
the engines instead.

Every tensor is returned in bf16 (the reference casts checkpoints to bf16 at load,
model_configuration.py:1129-1132), so an fp32 oracle fed ``w.float()`` and the bf16 HIP path
see bit-identical parameter values.
"""
import math
from typing import Dict

import torch

from .config import DiTConfig, VAEConfig

SEED_WEIGHTS = 1234


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


class _Maker:
    """device "meta": shapes only (no generator, no memory) -- how checkpoint.py enumerates the expected key set."""

    def __init__(self, device, seed, dtype=torch.bfloat16):
        self.device = torch.device(device)
        self.g = None if self.device.type == "meta" else _gen(self.device, seed)
        self.dtype = dtype

    def normal(self, shape, std, mean=0.0):
        if self.g is None:
            return torch.empty(shape, device="meta", dtype=self.dtype)
        t = torch.randn(shape, generator=self.g, device=self.device, dtype=torch.float32)
        return (t * std + mean).to(self.dtype)

    def linear_w(self, out_f, in_f):
        return self.normal((out_f, in_f), 1.0 / math.sqrt(in_f))

    def bias(self, n, std=0.02):
        return self.normal((n,), std)

    def gain(self, n, std=0.1):
        return self.normal((n,), std, mean=1.0)


def rope_freqs_lang(dim: int = 42, theta: float = 10000.0) -> torch.Tensor:
    """rotary_embedding_torch ``freqs_for='lang'``: 1/theta^(arange(0,dim,2)[:dim//2]/dim).
    NaMMRotaryEmbedding3d uses dim = 128 // 3 = 42 -> 21 freqs (dit_3b/rope.py:76-81)."""
    return 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))


def rope_freqs_pixel(dim: int = 21, max_freq: float = 256.0) -> torch.Tensor:
    """rotary_embedding_torch ``freqs_for='pixel'``: linspace(1, max_freq / 2, dim // 2) * pi.
    The 7B NaRotaryEmbedding3d uses dim = (128 // 2) // 3 = 21 -> 10 freqs (dit_7b/rope.py:28-33, mmsr_block.py:64)."""
    return torch.linspace(1.0, max_freq / 2, dim // 2) * math.pi


def synth_dit_state_dict(cfg: DiTConfig, device="cpu", seed: int = SEED_WEIGHTS) -> Dict[str, torch.Tensor]:
    mk = _Maker(device, seed)
    d, hd = cfg.vid_dim, cfg.head_dim
    inner = cfg.heads * hd
    hid = cfg.mlp_hidden
    sd: Dict[str, torch.Tensor] = {}
    sd["vid_in.proj.weight"] = mk.linear_w(d, cfg.patch_in_dim)
    sd["vid_in.proj.bias"] = mk.bias(d)
    sd["txt_in.weight"] = mk.linear_w(d, cfg.txt_in_dim)
    sd["txt_in.bias"] = mk.bias(d)
    sd["emb_in.proj_in.weight"] = mk.linear_w(d, 256)
    sd["emb_in.proj_in.bias"] = mk.bias(d)
    sd["emb_in.proj_hid.weight"] = mk.linear_w(d, d)
    sd["emb_in.proj_hid.bias"] = mk.bias(d)
    sd["emb_in.proj_out.weight"] = mk.linear_w(cfg.emb_dim, d)
    sd["emb_in.proj_out.bias"] = mk.bias(cfg.emb_dim)
    for i in range(cfg.num_layers):
        branches = ("vid", "txt") if i < cfg.mm_layers else ("all",)
        p = f"blocks.{i}."
        for b in branches:
            sd[p + f"attn.proj_qkv.{b}.weight"] = mk.linear_w(3 * inner, d)
        for b in branches:
            sd[p + f"attn.proj_out.{b}.weight"] = mk.linear_w(d, inner)
            sd[p + f"attn.proj_out.{b}.bias"] = mk.bias(d)
        for b in branches:
            sd[p + f"attn.norm_q.{b}.weight"] = mk.gain(hd)
        for b in branches:
            sd[p + f"attn.norm_k.{b}.weight"] = mk.gain(hd)
        if cfg.rope_type == "rope3d":
            sd[p + "attn.rope.rope.freqs"] = rope_freqs_pixel((hd // 2) // 3).to(mk.device)
        else:
            sd[p + "attn.rope.rope.freqs"] = rope_freqs_lang(cfg.rope_dim // 3).to(mk.device)
        for b in branches:
            if cfg.mlp_type == "normal":
                sd[p + f"mlp.{b}.proj_in.weight"] = mk.linear_w(hid, d)
                sd[p + f"mlp.{b}.proj_in.bias"] = mk.bias(hid)
                sd[p + f"mlp.{b}.proj_out.weight"] = mk.linear_w(d, hid)
                sd[p + f"mlp.{b}.proj_out.bias"] = mk.bias(d)
            else:
                sd[p + f"mlp.{b}.proj_in_gate.weight"] = mk.linear_w(hid, d)
                sd[p + f"mlp.{b}.proj_out.weight"] = mk.linear_w(d, hid)
                sd[p + f"mlp.{b}.proj_in.weight"] = mk.linear_w(hid, d)
        for b in branches:
            for layer in ("attn", "mlp"):
                # AdaSingle init, dit_3b/modulation.py:58-63
                sd[p + f"ada.{b}.{layer}_shift"] = mk.normal((d,), 1.0 / math.sqrt(d))
                sd[p + f"ada.{b}.{layer}_scale"] = mk.normal((d,), 1.0 / math.sqrt(d), mean=1.0)
                sd[p + f"ada.{b}.{layer}_gate"] = mk.normal((d,), 1.0 / math.sqrt(d))
    if cfg.out_norm:
        sd["vid_out_norm.weight"] = mk.gain(d)
        sd["vid_out_ada.out_shift"] = mk.normal((d,), 1.0 / math.sqrt(d))
        sd["vid_out_ada.out_scale"] = mk.normal((d,), 1.0 / math.sqrt(d), mean=1.0)
    sd["vid_out.proj.weight"] = mk.linear_w(cfg.patch_out_dim, d)
    sd["vid_out.proj.bias"] = mk.bias(cfg.patch_out_dim)
    return sd


def _conv(mk: _Maker, sd, name, cout, cin, k):
    kt, kh, kw = k
    sd[name + ".weight"] = mk.normal((cout, cin, kt, kh, kw), 1.0 / math.sqrt(cin * kt * kh * kw))
    sd[name + ".bias"] = mk.bias(cout)


def _gn(mk: _Maker, sd, name, c):
    sd[name + ".weight"] = mk.gain(c)
    sd[name + ".bias"] = mk.bias(c, std=0.1)


def _resnet(mk, sd, name, cin, cout):
    _gn(mk, sd, name + ".norm1", cin)
    _conv(mk, sd, name + ".conv1", cout, cin, (3, 3, 3))
    _gn(mk, sd, name + ".norm2", cout)
    _conv(mk, sd, name + ".conv2", cout, cout, (3, 3, 3))
    if cin != cout:
        _conv(mk, sd, name + ".conv_shortcut", cout, cin, (1, 1, 1))


def _mid(mk, sd, name, c):
    # the reference registers attentions before resnets (attn_video_vae.py:648-649)
    a = name + ".attentions.0"
    _gn(mk, sd, a + ".group_norm", c)
    for lin in ("to_q", "to_k", "to_v", "to_out.0"):
        sd[f"{a}.{lin}.weight"] = mk.linear_w(c, c)
        sd[f"{a}.{lin}.bias"] = mk.bias(c)
    _resnet(mk, sd, name + ".resnets.0", c, c)
    _resnet(mk, sd, name + ".resnets.1", c, c)


def synth_vae_state_dict(cfg: VAEConfig, device="cpu", seed: int = SEED_WEIGHTS + 1) -> Dict[str, torch.Tensor]:
    """Keys follow the diffusers-style layout the reference VAE registers
    (attn_video_vae.py Encoder3D :671 / Decoder3D :859)."""
    mk = _Maker(device, seed)
    ch = cfg.block_out_channels
    n = len(ch)
    sd: Dict[str, torch.Tensor] = {}
    # ---- encoder
    _conv(mk, sd, "encoder.conv_in", ch[0], cfg.in_channels, (3, 3, 3))
    cout = ch[0]
    for i in range(n):
        cin, cout = cout, ch[i]
        for j in range(cfg.layers_per_block):
            _resnet(mk, sd, f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != n - 1:
            temporal = i >= n - cfg.temporal_scale_num - 1      # attn_video_vae.py:744
            k = (3, 3, 3) if temporal else (1, 3, 3)
            _conv(mk, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, k)
    _mid(mk, sd, "encoder.mid_block", ch[-1])
    _gn(mk, sd, "encoder.conv_norm_out", ch[-1])
    _conv(mk, sd, "encoder.conv_out", 2 * cfg.latent_channels, ch[-1], (3, 3, 3))
    # ---- decoder
    _conv(mk, sd, "decoder.conv_in", ch[-1], cfg.latent_channels, (3, 3, 3))
    rev = list(reversed(ch))
    cout = rev[0]
    for i in range(n):
        cin, cout = cout, rev[i]
        for j in range(cfg.layers_per_block + 1):
            _resnet(mk, sd, f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != n - 1:
            temporal = i < cfg.temporal_scale_num                 # attn_video_vae.py:944
            ratio = 4 * (2 if temporal else 1)
            u = f"decoder.up_blocks.{i}.upsamplers.0"
            _conv(mk, sd, u + ".conv", cout, cout, (3, 3, 3))
            _conv(mk, sd, u + ".upscale_conv", cout * ratio, cout, (1, 1, 1))
    _mid(mk, sd, "decoder.mid_block", ch[-1])
    _gn(mk, sd, "decoder.conv_norm_out", ch[0])
    _conv(mk, sd, "decoder.conv_out", cfg.out_channels, ch[0], (3, 3, 3))
    return sd


def synth_text_embedding(n_tokens: int = 58, dim: int = 5120, device="cpu", seed: int = 7) -> torch.Tensor:
    """Stand-in for the reference's shipped ``pos_emb.pt`` ([58, 5120] bf16, std 0.975)."""
    g = _gen(torch.device(device), seed)
    return (torch.randn((n_tokens, dim), generator=g, device=device) * 0.975).to(torch.bfloat16)
