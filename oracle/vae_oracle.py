"""TEST INFRASTRUCTURE -- CPU restatement of the reference causal-Conv3d video VAE (v3, s8/t4/c16).

Functional torch implementation driven by a reference-named state dict; NCDHW, any float
dtype.  Temporal slicing in the reference is result-equivalent to the un-sliced causal conv
(verified there to ~1e-6), so the oracle runs un-sliced; spatial tiling
*does* change results (per-tile GroupNorm / attention) and is restated tile for tile.

Reference functions restated (file:line under /root/reference/src/models/video_vae_v3/modules):
  InflatedCausalConv3d.forward / extend_head     causal_inflation_lib.py:213-248, 422-437
  causal_norm_wrapper (per-frame GroupNorm)      causal_inflation_lib.py:354-409
  ResnetBlock3D.forward                          attn_video_vae.py:311-362
  Downsample3D.forward / Upsample3D.forward      attn_video_vae.py:229-250 / 110-174
  UNetMidBlock3D.forward (+ diffusers Attention) attn_video_vae.py:656-668, 615-632
  Encoder3D.forward / Decoder3D.forward          attn_video_vae.py:808-856 / 983-1035
  tiled_encode / tiled_decode                    attn_video_vae.py:1302-1468 / 1470-1630
  VideoDiffusionInfer.vae_encode / vae_decode    src/core/infer.py:117-199 / 203-278
"""
import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


def causal_conv3d(x, sd, name, dtype, stride=(1, 1, 1), spatial_pad=(1, 1)):
    w = sd[name + ".weight"].to(dtype)
    b = sd[name + ".bias"].to(dtype)
    kt = w.shape[2]
    head = kt - 1                      # temporal_padding * 2, first frame replicated
    if head > 0:
        x = torch.cat([x[:, :, :1]] * head + [x], dim=2)
    return F.conv3d(x, w, b, stride=stride, padding=(0, spatial_pad[0], spatial_pad[1]))


def frame_group_norm(x, sd, name, dtype, groups, eps):
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.group_norm(y, groups, sd[name + ".weight"].to(dtype), sd[name + ".bias"].to(dtype), eps)
    return y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)


def resnet(x, sd, name, dtype, cfg):
    h = F.silu(frame_group_norm(x, sd, name + ".norm1", dtype, cfg.norm_num_groups, cfg.norm_eps))
    h = causal_conv3d(h, sd, name + ".conv1", dtype)
    h = F.silu(frame_group_norm(h, sd, name + ".norm2", dtype, cfg.norm_num_groups, cfg.norm_eps))
    h = causal_conv3d(h, sd, name + ".conv2", dtype)
    if (name + ".conv_shortcut.weight") in sd:
        x = causal_conv3d(x, sd, name + ".conv_shortcut", dtype, spatial_pad=(0, 0))
    return x + h


def mid_block(x, sd, name, dtype, cfg):
    x = resnet(x, sd, name + ".resnets.0", dtype, cfg)
    b, c, t, h, w = x.shape
    a = name + ".attentions.0"
    res = x
    y = frame_group_norm(x, sd, a + ".group_norm", dtype, cfg.norm_num_groups, cfg.norm_eps)
    y = y.permute(0, 2, 3, 4, 1).reshape(b * t, h * w, c)          # per-frame tokens
    q = F.linear(y, sd[a + ".to_q.weight"].to(dtype), sd[a + ".to_q.bias"].to(dtype))
    k = F.linear(y, sd[a + ".to_k.weight"].to(dtype), sd[a + ".to_k.bias"].to(dtype))
    v = F.linear(y, sd[a + ".to_v.weight"].to(dtype), sd[a + ".to_v.bias"].to(dtype))
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(c), dim=-1)
    o = F.linear(p @ v, sd[a + ".to_out.0.weight"].to(dtype), sd[a + ".to_out.0.bias"].to(dtype))
    o = o.reshape(b, t, h, w, c).permute(0, 4, 1, 2, 3)
    x = o + res
    return resnet(x, sd, name + ".resnets.1", dtype, cfg)


def encoder(x, sd, cfg, dtype):
    ch = cfg.block_out_channels
    n = len(ch)
    x = causal_conv3d(x, sd, "encoder.conv_in", dtype)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            x = resnet(x, sd, f"encoder.down_blocks.{i}.resnets.{j}", dtype, cfg)
        if i != n - 1:
            temporal = i >= n - cfg.temporal_scale_num - 1
            x = F.pad(x, (0, 1, 0, 1))                                  # attn_video_vae.py:242-244
            x = causal_conv3d(x, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", dtype,
                              stride=(2 if temporal else 1, 2, 2), spatial_pad=(0, 0))
    x = mid_block(x, sd, "encoder.mid_block", dtype, cfg)
    x = F.silu(frame_group_norm(x, sd, "encoder.conv_norm_out", dtype, cfg.norm_num_groups, cfg.norm_eps))
    return causal_conv3d(x, sd, "encoder.conv_out", dtype)


def upsample(x, sd, name, dtype, temporal):
    w = sd[name + ".upscale_conv.weight"].to(dtype)
    b = sd[name + ".upscale_conv.bias"].to(dtype)
    y = F.conv3d(x, w, b)
    B, CC, f, h, ww = y.shape
    rz = 2 if temporal else 1
    C = CC // (4 * rz)
    # "b (x y z c) f h w -> b c (f z) (h x) (w y)"
    y = y.reshape(B, 2, 2, rz, C, f, h, ww).permute(0, 4, 5, 3, 6, 1, 7, 2).reshape(B, C, f * rz, h * 2, ww * 2)
    if temporal:
        y = torch.cat([y[:, :, :1], y[:, :, 2:]], dim=2)             # remove_head(times=1)
    return causal_conv3d(y, sd, name + ".conv", dtype)


def decoder(z, sd, cfg, dtype):
    ch = cfg.block_out_channels
    n = len(ch)
    x = causal_conv3d(z, sd, "decoder.conv_in", dtype)
    x = mid_block(x, sd, "decoder.mid_block", dtype, cfg)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = resnet(x, sd, f"decoder.up_blocks.{i}.resnets.{j}", dtype, cfg)
        if i != n - 1:
            x = upsample(x, sd, f"decoder.up_blocks.{i}.upsamplers.0", dtype, temporal=i < cfg.temporal_scale_num)
    x = F.silu(frame_group_norm(x, sd, "decoder.conv_norm_out", dtype, cfg.norm_num_groups, cfg.norm_eps))
    return causal_conv3d(x, sd, "decoder.conv_out", dtype)


def _ramps(n, dtype):
    if n <= 0:
        return None
    return 0.5 - 0.5 * torch.cos(torch.linspace(0, 1, steps=n, dtype=dtype) * math.pi)


def _tile_starts(total, tile, overlap):
    stride = max(1, tile - overlap)
    out = []
    for s in range(0, total, stride):
        e = min(s + tile, total)
        if s > 0 and (e - s) <= overlap:
            continue
        out.append((s, e))
    return out


def _edge_weight(length, ov, ramp, fade_lo, fade_hi, dtype):
    wgt = torch.ones(length, dtype=dtype)
    ov = max(0, min(ov, length - 1))
    if ov > 0:
        if fade_lo:
            wgt[:ov] = ramp[:ov]
        if fade_hi:
            wgt[-ov:] = 1 - ramp[:ov]
    return wgt


def encode(x, sd, cfg, dtype=torch.float32, tiled=False, tile_size=(512, 512), tile_overlap=(64, 64)):
    """x [B, 3, T, H, W] -> posterior mean [B, 16, T', H/8, W/8]."""
    x = x.to(dtype)
    B, _, T, H, W = x.shape
    s = cfg.spatial_downsample_factor
    if not tiled or (H <= tile_size[0] and W <= tile_size[1]):
        return encoder(x, sd, cfg, dtype)[:, : cfg.latent_channels]
    lth, ltw = max(1, tile_size[0] // s), max(1, tile_size[1] // s)
    loh = max(0, min(tile_overlap[0] // s, lth - 1))
    low = max(0, min(tile_overlap[1] // s, ltw - 1))
    Hl, Wl = (H + s - 1) // s, (W + s - 1) // s
    rh, rw = _ramps(loh, dtype), _ramps(low, dtype)
    result = count = None
    for (y0, y1) in _tile_starts(Hl, lth, loh):
        for (x0, x1) in _tile_starts(Wl, ltw, low):
            tile = encoder(x[:, :, :, y0 * s:min(y1 * s, H), x0 * s:min(x1 * s, W)], sd, cfg, dtype)
            if result is None:
                result = torch.zeros(B, tile.shape[1], tile.shape[2], Hl, Wl, dtype=dtype)
                count = torch.zeros(1, 1, 1, Hl, Wl, dtype=dtype)
            eh = min(y1 - y0, tile.shape[3], Hl - y0)
            ew = min(x1 - x0, tile.shape[4], Wl - x0)
            tile = tile[:, :, :, :eh, :ew]
            wh = _edge_weight(eh, loh, rh, y0 > 0, y1 < Hl, dtype).view(1, 1, 1, eh, 1)
            ww = _edge_weight(ew, low, rw, x0 > 0, x1 < Wl, dtype).view(1, 1, 1, 1, ew)
            result[:, :, :, y0:y0 + eh, x0:x0 + ew] += tile * wh * ww
            count[:, :, :, y0:y0 + eh, x0:x0 + ew] += wh * ww
    result = result / count.clamp(min=1e-6)
    return result[:, : cfg.latent_channels]


def decode(z, sd, cfg, dtype=torch.float32, tiled=False, tile_size=(512, 512), tile_overlap=(64, 64)):
    """z [B, 16, T', h, w] -> sample [B, 3, T, 8h, 8w]."""
    z = z.to(dtype)
    B, _, Tl, H, W = z.shape
    s = cfg.spatial_downsample_factor
    lth, ltw = max(1, tile_size[0] // s), max(1, tile_size[1] // s)
    if not tiled or (H <= lth and W <= ltw):
        return decoder(z, sd, cfg, dtype)
    oh, ow = tile_overlap
    loh = max(0, min(oh // s, lth - 1))
    low = max(0, min(ow // s, ltw - 1))
    rh, rw = _ramps(oh, dtype), _ramps(ow, dtype)       # decode ramps live in OUTPUT pixels
    result = count = None
    for (y0, y1) in _tile_starts(H, lth, loh):
        for (x0, x1) in _tile_starts(W, ltw, low):
            tile = decoder(z[:, :, :, y0:y1, x0:x1], sd, cfg, dtype)
            if result is None:
                result = torch.zeros(B, tile.shape[1], tile.shape[2], H * s, W * s, dtype=dtype)
                count = torch.zeros(1, 1, 1, H * s, W * s, dtype=dtype)
            ho, wo = (y1 - y0) * s, (x1 - x0) * s
            wh = _edge_weight(ho, oh, rh, y0 > 0, y1 < H, dtype).view(1, 1, 1, ho, 1)
            ww = _edge_weight(wo, ow, rw, x0 > 0, x1 < W, dtype).view(1, 1, 1, 1, wo)
            result[:, :, :, y0 * s:y1 * s, x0 * s:x1 * s] += tile * wh * ww
            count[:, :, :, y0 * s:y1 * s, x0 * s:x1 * s] += wh * ww
    return result / count.clamp(min=1e-6)


def runner_vae_encode(sample_cthw, sd, cfg, dtype=torch.float32, **tile_kw):
    """infer.py:117-199 for one sample: [3, T, H, W] -> latent [T', H/8, W/8, 16] (scaled)."""
    lat = encode(sample_cthw.unsqueeze(0), sd, cfg, dtype, **tile_kw)
    lat = lat.permute(0, 2, 3, 4, 1)
    return ((lat - cfg.shifting_factor) * cfg.scaling_factor).squeeze(0)


def runner_vae_decode(latent_thwc, sd, cfg, dtype=torch.float32, **tile_kw):
    """infer.py:203-278 for one latent: [T', h, w, 16] -> sample [3, T, H, W]."""
    z = (latent_thwc.to(dtype) / cfg.scaling_factor + cfg.shifting_factor).permute(3, 0, 1, 2).unsqueeze(0)
    return decode(z, sd, cfg, dtype, **tile_kw).squeeze(0)
