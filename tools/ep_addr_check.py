#!/usr/bin/env python
"""CPU check of the address algebra behind -DSVR_EP_ADDR (csrc/svr_conv_halo2.hip, csrc/svr_conv_sub.hip): for random ragged and
interior tile geometries, every store slot (thread, epilogue pass, slot q) of the LDS-staged epilogue must get the same in-image mask,
the same border code (sub-pixel conv) and the same element offset from the split form -- wave-uniform row term per parked row + one of
two per-thread column terms -- as from the shipped per-chunk formula.  Restates both forms in Python; no GPU, no library.
    python tools/ep_addr_check.py [trials]
"""
import random
import sys

NT, MTW = 256, 8


def halo_shipped(to, H, W, y0, x0, n0, ld, tid, p, q):
    vox = (q * NT + tid) >> 4
    r = vox >> 5
    y, x = y0 + (r >> 1) * MTW + 2 * p + (r & 1), x0 + (vox & 31)
    return (y < H and x < W), ((to * H + min(y, H - 1)) * W + min(x, W - 1)) * ld + n0 + (tid & 15) * 8


def halo_split(to, H, W, y0, x0, n0, ld, tid, p, q):
    kmax, row0, wld = H - 1 - y0, to * H + y0, W * ld
    j, rq = q & 1, q >> 1
    xq = x0 + j * 16 + (tid >> 4)
    col = min(xq, W - 1) * ld + n0 + (tid & 15) * 8
    k = (rq >> 1) * MTW + 2 * p + (rq & 1)
    return (k <= kmax and xq < W), (row0 + min(k, kmax)) * wld + col


def sub_shipped(to, ts, up, py, px, H, W, y0, x0, n, N, tid, p, q, yb, xb):
    vox = (q * NT + tid) >> 4
    r = vox >> 5
    y, x = y0 + (r >> 1) * MTW + 2 * p + (r & 1), x0 + (vox & 31)
    yc, xc = min(y, H - 1), min(x, W - 1)
    return (y < H and x < W), (1 if yc == yb else 0) | (2 if xc == xb else 0), \
        ((to * ts * (up * H) + up * yc + py) * (up * W) + up * xc + px) * N + n


def sub_split(to, ts, up, py, px, H, W, y0, x0, n, N, tid, p, q, yb, xb):
    kmax, rowbase, pitch = H - 1 - y0, to * ts * (up * H) + up * y0 + py, (up * W) * N
    j, rq = q & 1, q >> 1
    xq = x0 + j * 16 + (tid >> 4)
    xc = min(xq, W - 1)
    k = (rq >> 1) * MTW + 2 * p + (rq & 1)
    kk = min(k, kmax)
    return (k <= kmax and xq < W), (1 if y0 + kk == yb else 0) | (2 if xc == xb else 0), (rowbase + up * kk) * pitch + (up * xc + px) * N + n


def main(trials=300):
    random.seed(1)
    slots = 0
    for _ in range(trials):
        H, W = random.choice([16, 17, 31, 40, 200, 1000, 1024]), random.choice([32, 33, 40, 63, 328, 1000, 1024])
        y0, x0 = 16 * random.randrange((H + 15) // 16), 32 * random.randrange((W + 31) // 32)
        to, n0, ld = random.randrange(5), 128 * random.randrange(4), random.choice([128, 256, 512, 640])
        for tid in random.sample(range(256), 24):
            for p in range(4):
                for q in range(8):
                    assert halo_shipped(to, H, W, y0, x0, n0, ld, tid, p, q) == halo_split(to, H, W, y0, x0, n0, ld, tid, p, q)
                    for py, px in ((0, 0), (0, 1), (1, 0), (1, 1)):
                        for ts, up in ((1, 2), (2, 2), (1, 1)):
                            a = (to, ts, up, py if up == 2 else 0, px if up == 2 else 0, H, W, y0, x0, n0 + (tid & 15) * 8, ld, tid, p, q,
                                 H - 1 if py else 0, W - 1 if px else 0)
                            assert sub_shipped(*a) == sub_split(*a)
                    slots += 1
    print(f"{slots} store slots: identical masks, border codes and element offsets")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
