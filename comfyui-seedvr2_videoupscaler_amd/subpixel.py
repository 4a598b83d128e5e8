"""Sub-pixel form of the VAE's spatial upsampler.

``Upsample3D`` (attn_video_vae.py:110-174) is ``upscale_conv`` (1x1x1, C -> 4C) -> pixel shuffle
``"b (x y z c) f h w -> b c (f z) (h x) (w y)"`` (:137-143) -> causal 3x3x3 conv.  Both steps are linear, so for a
spatial-only upsampler (temporal ratio 1) every output phase (py, px) of the 2x grid is ONE conv over the LOW-resolution
input with a (kt, 2, 2) window: output row 2h + py reads the upsampled rows 2h + py - 1 .. 2h + py + 1, i.e. the
low-resolution rows {h - 1, h} (py = 0) or {h, h + 1} (py = 1), each through the sub-row of ``upscale_conv`` it came from.
Merged weights: ``Wm[py, px][:, :, dt, ry, rx] = sum over the (ky, kx) that land on window position (ry, rx) of
W3[:, :, dt, ky, kx] @ W1[block(ky, kx)]`` -- 12 instead of 28 multiply-adds per output voxel and channel pair, and the
upsampled intermediate never exists.  The zero padding of the upsampled grid becomes the zero padding of the
low-resolution input for the data term; the ``upscale_conv`` BIAS term of a padded tap must vanish too, hence a separate
bias vector for the voxels on the image border (row, column, corner).  Exact in real arithmetic
(tests/test_host_logic.py::test_subpixel_merge_matches_the_two_step_upsampler); in bf16 it rounds the merged weights once
instead of rounding the upsampled intermediate.
"""
from typing import List, Tuple

import torch


def tap_map(p: int, k: int) -> Tuple[int, int]:
    """Output phase p in {0, 1}, tap k in {0, 1, 2} of the 3-tap window on the upsampled axis ->
    (window position 0 | 1 on the low-resolution axis, sub-position 0 | 1 inside that low-resolution voxel)."""
    d = p + k - 1
    return d // 2 - (-1 if p == 0 else 0), d % 2


def merge_spatial_upsampler(w1: torch.Tensor, b1: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor
                            ) -> List[Tuple[int, int, torch.Tensor, torch.Tensor, torch.Tensor]]:
    """w1 [4C, C] (rows ordered (x y c): x = row sub-position, y = column sub-position), b1 [4C], w3 [Cout, C, kt, 3, 3],
    b3 [Cout] -> [(py, px, w [Cout, C, kt, 2, 2], bias [Cout], bias_border [3, Cout])] in fp32; ``bias_border`` rows: voxel on
    the row border (y == 0 for py = 0, y == H - 1 for py = 1), on the column border, on both."""
    c = w1.shape[1]
    assert w1.shape[0] == 4 * c and w3.shape[1] == c and tuple(w3.shape[3:]) == (3, 3)
    w1, b1, w3, b3 = (t.float() for t in (w1, b1, w3, b3))
    cout, kt = w3.shape[0], w3.shape[2]
    out = []
    for py in range(2):
        for px in range(2):
            wm = torch.zeros(cout, c, kt, 2, 2, dtype=torch.float32, device=w3.device)
            bias = [b3.clone() for _ in range(4)]                    # interior, row border, column border, corner
            for ky in range(3):
                ry, ys = tap_map(py, ky)
                row_tap_is_padding_on_the_border = (py == 0 and ky == 0) or (py == 1 and ky == 2)
                for kx in range(3):
                    rx, xs = tap_map(px, kx)
                    col_tap_is_padding_on_the_border = (px == 0 and kx == 0) or (px == 1 and kx == 2)
                    blk = ys * 2 + xs
                    wb, bb = w1[blk * c:(blk + 1) * c], b1[blk * c:(blk + 1) * c]
                    w3t = w3[:, :, :, ky, kx]                                        # [Cout, C, kt]
                    wm[:, :, :, ry, rx] += torch.einsum("omt,mi->oit", w3t, wb)
                    contrib = torch.einsum("omt,m->o", w3t, bb)
                    bias[0] += contrib
                    if not row_tap_is_padding_on_the_border:
                        bias[1] += contrib
                    if not col_tap_is_padding_on_the_border:
                        bias[2] += contrib
                    if not (row_tap_is_padding_on_the_border or col_tap_is_padding_on_the_border):
                        bias[3] += contrib
            out.append((py, px, wm, bias[0], torch.stack(bias[1:], 0)))
    return out
