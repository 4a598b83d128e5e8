"""Sub-pixel form of the VAE's upsamplers.

``Upsample3D`` (attn_video_vae.py:110-174) is ``upscale_conv`` (1x1x1, C -> 4 rz C) -> pixel shuffle
``"b (x y z c) f h w -> b c (f z) (h x) (w y)"`` (:137-143) -> (temporal upsamplers, first slice: ``remove_head`` drops the
duplicated second frame, :152-153) -> causal 3x3x3 conv.  Every step is linear, so each output phase of the upsampled
grid is ONE small conv over the LOW-resolution input:

* space: output row 2h + py reads the upsampled rows 2h + py - 1 .. 2h + py + 1, i.e. the low-resolution rows {h - 1, h}
  (py = 0) or {h, h + 1} (py = 1), each through the sub-row of ``upscale_conv`` it came from -> a 2 x 2 window, four phases;
* time, rz = 1: the three causal taps are three low-resolution frames -> window (3, 2, 2), 12 instead of 28 multiply-adds per
  output voxel and channel pair;
* time, rz = 2: upsampled frame i >= 1 is low-resolution frame (i + 1) // 2, sub-frame (i + 1) % 2 (frame 0 keeps only its
  sub-frame 0), so the causal taps i - 2, i - 1, i (clamped at 0: the replicated head) touch at most TWO low-resolution frames
  -> window (2, 2, 2), 8 instead of 28.  The tap -> (frame, sub-frame) pattern (``signature``) has five values: i = 0, 1, 2
  (the head) and the two steady-state parities.

Merged weights: ``Wm[:, :, s, ry, rx] = sum over the taps (k, ky, kx) that land on window position (s, ry, rx) of
W3[:, :, k, ky, kx] @ W1[block(ky, kx, sub-frame)]``; the upsampled intermediate never exists.  The zero padding of the
upsampled grid becomes the zero padding of the low-resolution input for the data term; the ``upscale_conv`` BIAS term of a
padded tap must vanish too, hence a separate bias vector for the voxels on the image border (row, column, corner).  Exact in
real arithmetic (tests/test_host_logic.py); in bf16 it rounds the merged weights once instead of rounding the upsampled
intermediate.
"""
from typing import List, Tuple

import torch

Signature = Tuple[Tuple[int, int, int], ...]          # per causal tap k: (source frame relative to the output's, k, sub-frame)


def tap_map(p: int, k: int) -> Tuple[int, int]:
    """Output phase p in {0, 1}, tap k in {0, 1, 2} of the 3-tap window on the upsampled axis ->
    (window position 0 | 1 on the low-resolution axis, sub-position 0 | 1 inside that low-resolution voxel)."""
    d = p + k - 1
    return d // 2 - (-1 if p == 0 else 0), d % 2


def frame_of(i: int, rz: int) -> Tuple[int, int]:
    """Upsampled frame i of the whole clip -> (low-resolution frame, sub-frame)."""
    if rz == 1:
        return i, 0
    return (0, 0) if i == 0 else ((i + 1) // 2, (i + 1) % 2)


def signature(i: int, rz: int, kt: int = 3) -> Signature:
    t_out = frame_of(i, rz)[0]
    sig = []
    for k in range(kt):
        t, z = frame_of(max(i - (kt - 1) + k, 0), rz)
        sig.append((t - t_out, k, z))
    return tuple(sig)


def output_frames(t: int, rz: int) -> List[int]:
    """Upsampled frames that low-resolution frame t (index in the whole clip) produces."""
    if rz == 1:
        return [t]
    return [0] if t == 0 else [2 * t - 1, 2 * t]


def merge_upsampler(w1: torch.Tensor, b1: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, rz: int, sig: Signature):
    """w1 [4 rz C, C] (rows ordered (x y z c): x = row sub-position, y = column sub-position, z = sub-frame), b1 [4 rz C],
    w3 [Cout, C, kt, 3, 3], b3 [Cout] -> (sources, [(py, px, w [Cout, C, len(sources), 2, 2], bias [Cout], bias_border [3, Cout])])
    in fp32.  ``sources``: the distinct relative source frames of ``sig`` in ascending order = the temporal window;
    ``bias_border`` rows: voxel on the row border (y == 0 for py = 0, y == H - 1 for py = 1), on the column border, on both."""
    c = w1.shape[1]
    assert w1.shape[0] == 4 * rz * c and w3.shape[1] == c and tuple(w3.shape[3:]) == (3, 3) and len(sig) == w3.shape[2]
    w1, b1, w3, b3 = (t.float() for t in (w1, b1, w3, b3))
    cout = w3.shape[0]
    sources = sorted({s for s, _, _ in sig})
    out = []
    for py in range(2):
        for px in range(2):
            wm = torch.zeros(cout, c, len(sources), 2, 2, dtype=torch.float32, device=w3.device)
            bias = [b3.clone() for _ in range(4)]                    # interior, row border, column border, corner
            for s, k, z in sig:
                for ky in range(3):
                    ry, ys = tap_map(py, ky)
                    row_tap_is_padding_on_the_border = (py == 0 and ky == 0) or (py == 1 and ky == 2)
                    for kx in range(3):
                        rx, xs = tap_map(px, kx)
                        col_tap_is_padding_on_the_border = (px == 0 and kx == 0) or (px == 1 and kx == 2)
                        blk = (ys * 2 + xs) * rz + z
                        wb, bb = w1[blk * c:(blk + 1) * c], b1[blk * c:(blk + 1) * c]
                        w3t = w3[:, :, k, ky, kx]                                    # [Cout, C]
                        wm[:, :, sources.index(s), ry, rx] += w3t @ wb
                        contrib = w3t @ bb
                        bias[0] += contrib
                        if not row_tap_is_padding_on_the_border:
                            bias[1] += contrib
                        if not col_tap_is_padding_on_the_border:
                            bias[2] += contrib
                        if not (row_tap_is_padding_on_the_border or col_tap_is_padding_on_the_border):
                            bias[3] += contrib
            out.append((py, px, wm, bias[0], torch.stack(bias[1:], 0)))
    return sources, out
