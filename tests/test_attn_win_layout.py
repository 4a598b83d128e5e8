"""Lane-level emulation of the second-generation window-attention kernel (csrc/svr_attn_win.hip) on the CPU.

The kernel's correctness rests on index algebra: which 16-byte chunk every LDS-DMA lane fetches, which LDS address every
lane reads (ds_read_b128 / ds_read_b64_tr_b16), and how v_mfma_f32_32x32x16_bf16 distributes operands and results over
the 64 lanes.  This test restates exactly those formulas (same names as the kernel) in numpy on top of the documented
instruction semantics (cdna_hip_programming.md section 2 "ds_read_b64_tr_b16", section 3 "Fragment layout") and checks
that one (window, head) comes out as softmax(q k^T) v -- including a ragged last key tile, clamped query rows and the
16-byte widened output store.  It cannot prove the hardware semantics, only that the kernel is consistent with them;
the -m gpu tests do the rest.
"""
import numpy as np

KT, D, TILE = 64, 128, 64 * 128 * 2


def mfma_32x32x16(A, B, C):
    """A[lane] = 8 values: A[i = lane&31][k = 8*(lane>>5) + e]; B[lane]: B[k = 8*(lane>>5)+e][j = lane&31];
    C[lane][r]: row i = (r&3) + 8*(r>>2) + 4*(lane>>5), col j = lane&31."""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        Am[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = A[l]
        Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = B[l]
    Cm = Am @ Bm
    out = C.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += Cm[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def tr16_b64(lds, addr):
    """ds_read_b64_tr_b16: lane l (group g = l>>4, i = l&15) receives element j = the (i&3)-th bf16 of the 8 bytes at the
    address supplied by lane (g, 4*j + (i>>2))."""
    out = np.zeros((64, 4))
    for l in range(64):
        g, i = l >> 4, l & 15
        for j in range(4):
            src = g * 16 + 4 * j + (i >> 2)
            a = addr[src] + (i & 3) * 2
            assert addr[src] % 8 == 0
            out[l, j] = lds[a // 2]
    return out


def run_wave(q, k, v, L, q0, wave, scale, NW=4):
    """q,k,v: [L, 128] float (bf16-representable not required).  Returns {qpos: out row} for this wave's valid queries."""
    nk = (L + KT - 1) // KT
    lds = np.zeros(4 * TILE // 2)                        # element (2-byte) addressed
    lanes = np.arange(64)
    l31, hi = lanes & 31, lanes >> 5
    ka_ = [l31 * 256 + (((2 * ds + hi) ^ (l31 & 15)) << 4) for ds in range(8)]
    i16, G = lanes & 15, (lanes >> 4) & 1
    jr, c4 = i16 >> 2, i16 & 3
    va_ = [2 * TILE + hi * 1024 + jr * 256 + ((((m ^ jr) << 2) | (2 * G + (c4 >> 1))) << 4) + (c4 & 1) * 8 for m in range(4)]
    qpos = q0 + wave * 32 + l31
    qrow = np.minimum(qpos, L - 1)
    qf = [np.stack([q[qrow[l], 16 * ds + 8 * hi[l]:16 * ds + 8 * hi[l] + 8] for l in range(64)]) for ds in range(8)]
    o = [np.zeros((64, 16)) for _ in range(4)]
    m_run = np.full(64, -np.inf); l_run = np.zeros(64)
    c = scale * 1.4426950408889634

    def stage(t, buf):                                   # all NW waves' pieces
        for w in range(NW):
            for it in range(16 // NW):
                for lane in range(64):
                    st_key = w * 4 + (lane >> 4)
                    key = min(t * KT + it * 4 * NW + st_key, L - 1)
                    st_k = ((lane & 15) ^ (st_key & 15))
                    st_v = ((lane & 15) ^ ((lane >> 4) << 2))
                    dk = (buf * TILE + w * 1024 + it * NW * 1024 + lane * 16) // 2
                    dv = ((2 + buf) * TILE + w * 1024 + it * NW * 1024 + lane * 16) // 2
                    lds[dk:dk + 8] = k[key, st_k * 8:st_k * 8 + 8]
                    lds[dv:dv + 8] = v[key, st_v * 8:st_v * 8 + 8]

    stage(0, 0)
    off = 0
    for t in range(nk):
        if t + 1 < nk:
            stage(t + 1, (t & 1) ^ 1)
        sacc = [np.zeros((64, 16)), np.zeros((64, 16))]
        for kb in range(2):
            for ds in range(8):
                kf = np.stack([lds[(ka_[ds][l] + off + kb * 8192) // 2:(ka_[ds][l] + off + kb * 8192) // 2 + 8] for l in range(64)])
                sacc[kb] = mfma_32x32x16(kf, qf[ds], sacc[kb])
        if (t + 1) * KT > L:
            for kb in range(2):
                for r in range(16):
                    keyi = t * KT + 4 * hi + kb * 32 + (r & 3) + 8 * (r >> 2)
                    sacc[kb][keyi >= L, r] = -np.inf
        mx = np.maximum(sacc[0].max(1), sacc[1].max(1))
        mx = np.maximum(mx, mx[lanes ^ 32])
        m_new = np.maximum(m_run, mx)
        alpha = np.exp2(m_run * c - m_new * c)
        m_run = m_new
        pf = {}
        psum = np.zeros(64)
        for kb in range(2):
            for u in range(2):
                p = np.exp2(sacc[kb][:, 8 * u:8 * u + 8] * c - (m_new * c)[:, None])
                psum += p.sum(1)
                pf[kb, u] = p
        l_run = l_run * alpha + psum
        for m in range(4):
            o[m] *= alpha[:, None]
        for kb in range(2):
            for u in range(2):
                OFF = (2 * kb + u) * 4096
                for m in range(4):
                    v0 = tr16_b64(lds, va_[m] + off + OFF)
                    v1 = tr16_b64(lds, va_[m] + off + OFF + 2048)
                    o[m] = mfma_32x32x16(np.concatenate([v0, v1], 1), pf[kb, u], o[m])
        off += -TILE if (t & 1) else TILE
    l = l_run + l_run[lanes ^ 32]
    res = {}
    rows = {}
    for m in range(4):
        for rq in (0, 2):
            a = o[m][:, 4 * rq:4 * rq + 4] / l[:, None]
            b = o[m][:, 4 * rq + 4:4 * rq + 8] / l[:, None]
            # v_permlane32_swap(vdst = a, src = b): lanes 32-63 of a <-> lanes 0-31 of b
            a2, b2 = a.copy(), b.copy()
            a2[32:], b2[:32] = b[:32], a[32:]
            for lane in range(64):
                if qpos[lane] < L:
                    d0 = 32 * m + 8 * rq + 8 * hi[lane]
                    rows.setdefault(int(qpos[lane]), np.full(128, np.nan))[d0:d0 + 8] = np.concatenate([a2[lane], b2[lane]])
    return rows


import pytest


@pytest.mark.parametrize("NW", [4, 8])
def test_attn_win_index_algebra(NW):
    rng = np.random.default_rng(0)
    L = 150                                              # 3 key tiles, ragged tail (150 = 2*64 + 22), 2 query tiles
    q, k, v = (rng.standard_normal((L, D)) for _ in range(3))
    scale = 1 / np.sqrt(D)
    s = (q @ k.T) * scale
    p = np.exp(s - s.max(1, keepdims=True))
    want = (p / p.sum(1, keepdims=True)) @ v
    got = {}
    for q0, wave in (((0, 0), (0, 3), (128, 0)) if NW == 4 else ((0, 0), (0, 3), (0, 4))):
        got.update(run_wave(q, k, v, L, q0, wave, scale, NW))
    assert set(got) == set(range(0, 32)) | set(range(96, 128)) | set(range(128, 150))
    for r, row in got.items():
        assert not np.isnan(row).any()
        np.testing.assert_allclose(row, want[r], rtol=1e-9, atol=1e-9)


def test_attn_win_bank_model():
    """LDS bank model of MI355X_MICROARCH.md (LDS table): ds_read_b128 is served in 4 groups of 16 lanes, each must hit 16
    distinct 16-byte slots of the 256-byte bank row; ds_read_b64_tr_b16 in 2 groups of 32 lanes, each must cover 32
    distinct 8-byte slots."""
    lanes = np.arange(64)
    l31, hi = lanes & 31, lanes >> 5
    groups128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups128 += [[x + 32 for x in g] for g in groups128]
    for ds in range(8):
        addr = l31 * 256 + (((2 * ds + hi) ^ (l31 & 15)) << 4)
        for g in groups128:
            assert len({int(a // 16) % 16 for a in addr[g]}) == 16
    i16, G = lanes & 15, (lanes >> 4) & 1
    jr, c4 = i16 >> 2, i16 & 3
    for m in range(4):
        addr = hi * 1024 + jr * 256 + ((((m ^ jr) << 2) | (2 * G + (c4 >> 1))) << 4) + (c4 & 1) * 8
        for g in (range(0, 32), range(32, 64)):
            assert len({int(a // 8) % 32 for a in addr[list(g)]}) == 32
