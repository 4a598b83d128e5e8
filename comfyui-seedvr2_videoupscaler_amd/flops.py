"""Algorithmic FLOP counts of the hot path (multiply-add = 2 FLOP), SURVEY.md section 8(d):

  DiT   = L*N*(2*d*3d + 2*d*d + 3*2*d*h_mlp) + sum_layers sum_windows 4*H*dh*(len_w + Lt)^2
          + 2*N*132*d + 2*N*d*64                       (text-stream GEMMs / elementwise excluded)
  Conv  = 2*Cin*Cout*kt*kh*kw * To*Ho*Wo                per conv
  VAEattn = T*(4*C*n^2 + 8*C^2*n), n = h*w per frame

These are the figures ``roofline.achieved`` is computed from (padding FLOPs are NOT counted).
"""
from typing import Tuple

import numpy as np

from . import windows
from .config import DiTConfig, VAEConfig


def dit_flops(cfg: DiTConfig, grid: Tuple[int, int, int], Lt: int = 58) -> dict:
    t, h, w = grid
    N = t * h * w
    d, hm = cfg.vid_dim, cfg.mlp_hidden
    mlp = (2 if cfg.mlp_type == "normal" else 3) * 2 * d * hm          # 7B: two GEMMs of 4d; 3B: SwiGLU, three of 2.7d
    linear = cfg.num_layers * N * (2 * d * 3 * d + 2 * d * d + mlp)
    linear += 2 * N * cfg.patch_in_dim * d + 2 * N * d * cfg.patch_out_dim
    attn = 0
    for li in range(cfg.num_layers):
        plan = windows.plan_windows(grid, tuple(cfg.window), cfg.window_method(li))
        lens = np.diff(plan.cu).astype(np.float64) + Lt
        attn += float((4 * cfg.heads * cfg.head_dim * lens * lens).sum())
    return {"linear": float(linear), "attn": attn, "total": float(linear) + attn}


def conv_flops(cin, cout, k, out_vox) -> float:
    return 2.0 * cin * cout * k[0] * k[1] * k[2] * out_vox


def _c3(cin, cout, t, n, head=False):
    """Stride-1 3x3x3 causal conv over t frames of n voxels.  ``head``: as the engine runs it with ``merge_causal_head`` --
    output frame 0 of the clip with two temporal taps instead of three (vae.py:_conv_causal_head; not for thin inputs)."""
    f = conv_flops(cin, cout, (3, 3, 3), t * n)
    return f - conv_flops(cin, cout, (1, 3, 3), n) if head and cin >= 64 else f


def _resnet(cin, cout, t, n, head=False):
    f = _c3(cin, cout, t, n, head) + _c3(cout, cout, t, n, head)
    if cin != cout:
        f += conv_flops(cin, cout, (1, 1, 1), t * n)
    return f


def _mid(c, T, h, w, head=False):
    n = h * w
    return 2 * _resnet(c, c, T, n, head), T * (4.0 * c * n * n + 8.0 * c * c * n)


def vae_encode_flops(cfg: VAEConfig, T: int, H: int, W: int, causal_head: bool = False) -> dict:
    """``causal_head``: count frame 0 of every stride-1 3x3x3 conv as the engine runs it by default (see ``_c3``)."""
    ch = cfg.block_out_channels
    n = len(ch)
    conv = conv_flops(cfg.in_channels, ch[0], (3, 3, 3), T * H * W)
    t, h, w, c = T, H, W, ch[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            conv += _resnet(c if j == 0 else ch[i], ch[i], t, h * w, causal_head)
        c = ch[i]
        if i != n - 1:
            temporal = i >= n - cfg.temporal_scale_num - 1
            h, w = (h + 1 - 3) // 2 + 1, (w + 1 - 3) // 2 + 1
            if temporal:
                t = (t + 2 - 3) // 2 + 1
            conv += conv_flops(c, c, (3 if temporal else 1, 3, 3), t * h * w)
    m_conv, m_attn = _mid(c, t, h, w, causal_head)
    conv += m_conv + _c3(c, 2 * cfg.latent_channels, t, h * w, causal_head)
    return {"conv": conv, "attn": m_attn, "total": conv + m_attn}


def vae_decode_flops(cfg: VAEConfig, Tl: int, h: int, w: int, merged_upsamplers: bool = False,
                     causal_head: bool = False, keep_frames: int = None) -> dict:
    """``merged_upsamplers``: count the upsamplers as the engine runs them by default (sub-pixel convs over the
    low-resolution input, subpixel.py: 3 x 2 x 2 taps per output voxel for the spatial-only one, 2 x 2 x 2 for the temporal
    ones, no upscale_conv) instead of as the reference's upscale_conv + 3x3x3 conv.  ``causal_head``: see ``_c3``.
    ``keep_frames``: VideoVAEEngine.decode(keep_frames=): latents that only feed trimmed frames are dropped, and from the last
    temporal upsampler on only the kept frames are computed."""
    ch = list(reversed(cfg.block_out_channels))
    n = len(ch)
    tf = cfg.temporal_downsample_factor
    if keep_frames is not None and keep_frames < 1 + (Tl - 1) * tf:
        Tl = (keep_frames - 1 + tf - 1) // tf + 1
    else:
        keep_frames = None
    t, c = Tl, ch[0]
    conv = conv_flops(cfg.latent_channels, c, (3, 3, 3), t * h * w)
    m_conv, m_attn = _mid(c, t, h, w, causal_head)
    conv += m_conv
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            conv += _resnet(c if j == 0 else ch[i], ch[i], t, h * w, causal_head)
        c = ch[i]
        if i != n - 1:
            temporal = i < cfg.temporal_scale_num
            rz = 2 if temporal else 1
            if merged_upsamplers:                          # sub-pixel form: (3, 2, 2) / (2, 2, 2) taps per output voxel
                t, h, w = (t * 2 - 1 if temporal else t), h * 2, w * 2
                conv += conv_flops(c, c, (2 if temporal else 3, 2, 2), t * h * w)
                if temporal:                               # output frame 0 reads ONE low-resolution frame (subpixel.signature(0, 2))
                    conv -= conv_flops(c, c, (1, 2, 2), h * w)
                if keep_frames is not None and i == cfg.temporal_scale_num - 1:
                    t = min(t, keep_frames)                # (the upsampler itself still produced every frame of its slice)
                continue
            conv += conv_flops(c, c * 4 * rz, (1, 1, 1), t * h * w)
            t, h, w = (t * 2 - 1 if temporal else t), h * 2, w * 2
            conv += _c3(c, c, t, h * w, causal_head)
            if keep_frames is not None and i == cfg.temporal_scale_num - 1:
                t = min(t, keep_frames)
    conv += _c3(c, cfg.out_channels, t, h * w, causal_head)
    return {"conv": conv, "attn": m_attn, "total": conv + m_attn}


def _tiles(total, tile, overlap):
    stride = max(1, tile - overlap)
    out = []
    for s in range(0, total, stride):
        e = min(s + tile, total)
        if s > 0 and (e - s) <= overlap:
            continue
        out.append((s, e))
    return out


def vae_flops_tiled(cfg: VAEConfig, T: int, H: int, W: int, tiled: bool, tile=(1024, 1024), overlap=(128, 128),
                    merged_upsamplers: bool = False, causal_head: bool = False, keep_frames: int = None) -> dict:
    """Encode + decode FLOPs of one clip [T, H, W] (pixels), with the reference's tile grid if tiled."""
    s = cfg.spatial_downsample_factor
    Tl = (T - 1) // cfg.temporal_downsample_factor + 1
    Hl, Wl = (H + s - 1) // s, (W + s - 1) // s
    if not tiled or (H <= tile[0] and W <= tile[1]):
        return {"encode": vae_encode_flops(cfg, T, H, W, causal_head)["total"],
                "decode": vae_decode_flops(cfg, Tl, Hl, Wl, merged_upsamplers, causal_head, keep_frames)["total"]}
    lth, ltw = tile[0] // s, tile[1] // s
    loh, low = min(overlap[0] // s, lth - 1), min(overlap[1] // s, ltw - 1)
    enc = dec = 0.0
    for (y0, y1) in _tiles(Hl, lth, loh):
        for (x0, x1) in _tiles(Wl, ltw, low):
            enc += vae_encode_flops(cfg, T, min(y1 * s, H) - y0 * s, min(x1 * s, W) - x0 * s, causal_head)["total"]
            dec += vae_decode_flops(cfg, Tl, y1 - y0, x1 - x0, merged_upsamplers, causal_head, keep_frames)["total"]
    return {"encode": enc, "decode": dec}
