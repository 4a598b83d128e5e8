"""3D (shifted) window partition of the video token grid -- host-side integer bookkeeping.

Behaviour mirrored from the reference (results must be identical, element for element):
  make_720Pwindows_bysize / make_shifted_720Pwindows_bysize   src/models/dit_3b/window.py:28-83
  na.window_idx (gather / scatter index vectors)               src/models/dit_3b/na.py:616-641
  na.repeat_concat_idx (per-window [vid_w || txt] sequences)   src/models/dit_3b/na.py:320-424

Design here: each axis is cut independently into a list of [lo, hi) edges; the window set is
the cartesian product in the reference's order (w outermost, then h, then t innermost).  From
the boxes we build, once per (grid shape, method):
  * ``tok``   int32[sum L_w]  window-ordered position -> token row (t*H*W + h*W + w order)
  * ``pos``   int16[N, 3]     per token: window-local (f, h, w) coordinate (RoPE position)
  * ``cu``    int32[n_win+1]  cumulative video-token counts per window
so the HIP attention kernel can gather K/V rows and scatter outputs without ever
materialising a window-ordered copy of the activations.
"""
import math
from dataclasses import dataclass
from functools import lru_cache
from typing import List, Tuple

import numpy as np

REGULAR = "720pwin_by_size_bysize"
SHIFTED = "720pswin_by_size_bysize"


def _window_extent(t: int, h: int, w: int, num_windows: Tuple[int, int, int]) -> Tuple[int, int, int]:
    """Window size in tokens: the h x w grid is normalised to a 45x80 (=720p/16) area first."""
    nt, nh, nw = num_windows
    s = math.sqrt((45 * 80) / (h * w))
    rh, rw = round(h * s), round(w * s)          # python round (half-to-even), as the reference
    return math.ceil(min(t, 30) / nt), math.ceil(rh / nh), math.ceil(rw / nw)


def _axis_edges_regular(n: int, size: int) -> List[Tuple[int, int]]:
    out = []
    for i in range(math.ceil(n / size)):
        lo, hi = i * size, min((i + 1) * size, n)
        if hi > lo:
            out.append((lo, hi))
    return out


def _axis_edges_shifted(n: int, size: int) -> List[Tuple[int, int]]:
    shift = 0.5 if size < n else 0.0
    count = math.ceil((n - shift) / size)
    count = count + 1 if shift > 0 else 1
    out = []
    for i in range(count):
        lo = max(int((i - shift) * size), 0)
        hi = min(int((i - shift + 1) * size), n)
        if hi > lo:
            out.append((lo, hi))
    return out


def window_boxes(size: Tuple[int, int, int], num_windows: Tuple[int, int, int], method: str):
    """List of (t0, t1, h0, h1, w0, w1) boxes in the reference's enumeration order."""
    t, h, w = size
    wt, wh, ww = _window_extent(t, h, w, num_windows)
    if method == REGULAR:
        et, eh, ew = _axis_edges_regular(t, wt), _axis_edges_regular(h, wh), _axis_edges_regular(w, ww)
    elif method == SHIFTED:
        et, eh, ew = _axis_edges_shifted(t, wt), _axis_edges_shifted(h, wh), _axis_edges_shifted(w, ww)
    else:
        raise ValueError(f"Unknown windowing method: {method}")
    return [(a[0], a[1], b[0], b[1], c[0], c[1]) for c in ew for b in eh for a in et]


@dataclass(frozen=True)
class WindowPlan:
    """Index vectors for one (grid, method) pair.  All arrays are numpy, host side."""
    size: Tuple[int, int, int]
    n_win: int
    tok: np.ndarray        # int32 [N]     window-ordered -> token row
    pos: np.ndarray        # int16 [N, 3]  token row -> window-local (f, h, w)
    cu: np.ndarray         # int32 [n_win + 1]
    shapes: np.ndarray     # int32 [n_win, 3]
    max_len: int


@lru_cache(maxsize=64)
def plan_windows(size: Tuple[int, int, int], num_windows: Tuple[int, int, int], method: str) -> WindowPlan:
    t, h, w = size
    boxes = window_boxes(size, num_windows, method)
    tok_parts, lens, shapes = [], [], []
    pos = np.full((t * h * w, 3), -1, dtype=np.int16)
    grid = np.arange(t * h * w, dtype=np.int64).reshape(t, h, w)
    for (t0, t1, h0, h1, w0, w1) in boxes:
        rows = grid[t0:t1, h0:h1, w0:w1].reshape(-1)
        tok_parts.append(rows)
        lens.append(rows.size)
        shapes.append((t1 - t0, h1 - h0, w1 - w0))
        ff, hh, wwv = np.meshgrid(np.arange(t1 - t0), np.arange(h1 - h0), np.arange(w1 - w0), indexing="ij")
        pos[rows, 0] = ff.reshape(-1)
        pos[rows, 1] = hh.reshape(-1)
        pos[rows, 2] = wwv.reshape(-1)
    tok = np.concatenate(tok_parts).astype(np.int32)
    if tok.size != t * h * w or np.unique(tok).size != tok.size:
        raise AssertionError("window partition is not a permutation of the token grid")
    cu = np.zeros(len(lens) + 1, dtype=np.int32)
    cu[1:] = np.cumsum(lens)
    return WindowPlan(size=size, n_win=len(boxes), tok=tok, pos=pos, cu=cu,
                      shapes=np.asarray(shapes, dtype=np.int32), max_len=int(max(lens)))
