"""TEST INFRASTRUCTURE: a plain-PyTorch fp32 restatement of every op of the C ABI, with the same
Python signatures as ``<package>.ops.HipOps``.

Two uses, both in tests only (the product never imports this file):
  * -m gpu  : per-kernel numerics -- each HIP kernel is compared against the method of the same name;
  * -m "not gpu": host-logic tests -- the DiT / VAE engines are run on CPU with this backend injected,
    and compared with the oracle, so index plans, weight packing, temporal slicing, halos and tiling
    are verified without a GPU.
Inputs/outputs keep the product's storage dtypes (bf16 activations); arithmetic is fp32.
"""
import math

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16
H16, H16_SCALE = torch.float16, 2.0 ** -6      # the "h16" storage format of <package>.ops: IEEE half holding x * 2^-6


def _ld(t):
    """stored tensor -> fp32 values"""
    return t.float() * (1.0 / H16_SCALE) if t.dtype == H16 else t.float()


def _st(values, like):
    """fp32 values -> the storage format of tensor ``like``"""
    return (values * H16_SCALE).to(H16) if like.dtype == H16 else values.to(like.dtype)


EPI_BIAS, EPI_BIAS_SILU, EPI_RESID_GATE, EPI_SWIGLU, EPI_BIAS_GELU = 0, 1, 2, 3, 4


class TorchOps:
    name = "torch-reference"
    phase_quad = True

    def __init__(self, device="cpu", act_dtype=BF16):
        self.device = torch.device(device)
        self.act_dtype = act_dtype        # float32 -> exact host-logic checks, bf16 -> product storage regime

    def empty(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.act_dtype, device=self.device)

    # ------------------------------------------------------------------ GEMM / conv
    def gemm(self, A, W, out, *, N, K, M=None, bias=None, epilogue=EPI_BIAS, gate=None, resid=None,
             out_f32=False, conv=None, ps=None, lda=None, ldc=None, ldr=None, gn_groups=0, W_frag=None, phase=None):
        """``gn_groups`` > 0: return ``(out, None)`` like a HIP launch whose kernel cannot fuse the statistics."""
        if phase is not None and getattr(phase, "quad", None) is not None:       # the four phases of a quad launch, one by one
            import dataclasses
            for qpy, qpx, qw, qb, qbb, _ in phase.quad:
                one = type(phase)(qpy, qpx, qbb, phase.t_stride)
                self.gemm(A, qw, out, N=N, K=K, bias=qb, conv=dataclasses.replace(conv, pad=(conv.pad[0], 1 - qpy, 1 - qpx)),
                          phase=one, out_f32=out_f32)
            return (out, None) if gn_groups > 0 else out
        if gn_groups > 0:
            return self.gemm(A, W, out, N=N, K=K, M=M, bias=bias, epilogue=epilogue, gate=gate, resid=resid,
                             out_f32=out_f32, conv=conv, ps=ps, lda=lda, ldc=ldc, ldr=ldr, phase=phase), None
        Wf = W[:N, :K].float()
        if conv is not None:
            g = conv
            x = A.reshape(g.T, g.H, g.W, g.Cin).float()
            pt, ph, pw = g.pad
            kt, kh, kw = g.k
            st, sh, sw = g.stride
            if pt > 0:
                head = g.halo.float()[-pt:] if g.halo is not None else x[:1].expand(pt, g.H, g.W, g.Cin)
                x = torch.cat([head, x], dim=0)
            xin = x.permute(3, 0, 1, 2).unsqueeze(0)                            # [1, C, T, H, W]
            ph_hi = max(0, (g.Ho - 1) * sh + kh - g.H - ph)
            pw_hi = max(0, (g.Wo - 1) * sw + kw - g.W - pw)
            xin = F.pad(xin, (pw, pw_hi, ph, ph_hi))
            w5 = Wf[:, :kt * kh * kw * g.Cin].reshape(N, kt, kh, kw, g.Cin).permute(0, 4, 1, 2, 3)   # (thin: K zero-padded)
            y = F.conv3d(xin, w5, stride=(st, sh, sw))[0]                       # [N, To', Ho', Wo']
            y = y[:, :g.To, :g.Ho, :g.Wo]
            assert y.shape[1:] == (g.To, g.Ho, g.Wo), (y.shape, g)
            acc = y.permute(1, 2, 3, 0).reshape(-1, N)
            M = acc.shape[0]
        else:
            if M is None:
                M = A.shape[0]
            acc = A.reshape(-1, A.shape[-1])[:M, :K].float() @ Wf.t()
        if epilogue == EPI_SWIGLU:
            a4 = acc.reshape(M, N // 32, 2, 16)
            res = (F.silu(a4[:, :, 0]) * a4[:, :, 1]).reshape(M, N // 2)
        else:
            res = acc
            if bias is not None:
                res = res + bias[:N].float()
            if phase is not None and phase.bias_border is not None:      # border voxels take their own bias vector
                g = conv
                r4 = res.reshape(g.To, g.Ho, g.Wo, N).clone()
                a4 = acc.reshape(g.To, g.Ho, g.Wo, N)
                yb = g.Ho - 1 if phase.py else 0
                xb = g.Wo - 1 if phase.px else 0
                bb = phase.bias_border.float()
                r4[:, yb, :, :] = a4[:, yb, :, :] + bb[0]
                r4[:, :, xb, :] = a4[:, :, xb, :] + bb[1]
                r4[:, yb, xb, :] = a4[:, yb, xb, :] + bb[2]
                res = r4.reshape(-1, N)
            if epilogue == EPI_BIAS_GELU:
                res = F.gelu(res, approximate="tanh")
            elif epilogue == EPI_BIAS_SILU:
                res = F.silu(res)
            elif epilogue == EPI_RESID_GATE:
                if gate is not None:
                    res = res * gate[:N].float()
                if resid is not None:
                    res = res + _ld(resid.reshape(M, -1)[:, :N])
        if phase is not None:
            g = conv
            ts = getattr(phase, "t_stride", 1)
            o4 = out.reshape(-1, 2 * g.Ho, 2 * g.Wo, N)
            o4[0:(g.To - 1) * ts + 1:ts, phase.py::2, phase.px::2, :] = _st(res.reshape(g.To, g.Ho, g.Wo, N), out)
            return out
        if ps is not None:
            r = res.reshape(ps.F, ps.H, ps.W, 2, 2, ps.rz, ps.C).permute(0, 5, 1, 3, 2, 4, 6)
            r = r.reshape(ps.F * ps.rz, 2 * ps.H, 2 * ps.W, ps.C)
            if ps.drop_first:
                r = torch.cat([r[:1], r[2:]], dim=0)
            out.copy_(_st(r.reshape(out.shape), out))
            return out
        out.reshape(M, -1)[:, :res.shape[1]].copy_(_st(res, out))
        return out

    # ------------------------------------------------------------------ DiT side kernels
    def rmsnorm_mod(self, x, out, eps, w=None, scale=None, shift=None):
        xf = _ld(x)
        y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
        if w is not None:
            y = y * w
        if scale is not None:
            y = y * scale
        if shift is not None:
            y = y + shift
        out.copy_(y.to(out.dtype))
        return out

    def ada_combine(self, emb, params, slots, out):
        dim = params.shape[1]
        e = emb.float().reshape(dim, 6)
        out.copy_(e[:, slots.long()].t() + params.float())
        return out

    def qknorm_rope(self, qkv, heads, pos, t_offset, cos_tab, sin_tab, wq, wk, eps):
        rows = qkv.shape[0]
        v = qkv.float().reshape(rows, 3, heads, 128)
        nf = cos_tab.shape[1]
        p = pos.long().clone()
        p[:, 0] += t_offset
        cos = torch.cat([cos_tab[p[:, a]] for a in range(3)], dim=-1)           # [rows, 3*nf]
        sin = torch.cat([sin_tab[p[:, a]] for a in range(3)], dim=-1)
        for idx, wgt in ((0, wq), (1, wk)):
            t = v[:, idx]
            t = t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + eps) * wgt
            rot = t[..., : 6 * nf].reshape(rows, heads, 3 * nf, 2)
            x0, x1 = rot[..., 0], rot[..., 1]
            c, s = cos[:, None, :], sin[:, None, :]
            r = torch.stack((x0 * c - x1 * s, x1 * c + x0 * s), dim=-1).reshape(rows, heads, 6 * nf)
            v[:, idx] = torch.cat([r, t[..., 6 * nf:]], dim=-1)
        qkv.copy_(v.reshape(rows, -1).to(qkv.dtype))
        return qkv

    def softmax_rows(self, S, P, scale):
        P.copy_(torch.softmax(S.float() * scale, dim=-1).to(P.dtype))
        return P

    def attn_varlen(self, qkv, out, seq_rows, out_rows, cu, max_len, heads, head_dim, scale):
        q3 = qkv.float().reshape(qkv.shape[0], 3, heads, head_dim)
        cu_l = cu.tolist()
        for i in range(len(cu_l) - 1):
            src = seq_rows[cu_l[i]:cu_l[i + 1]].long()
            dst = out_rows[cu_l[i]:cu_l[i + 1]].long()
            q, k, v = (q3[src, j].transpose(0, 1) for j in range(3))                # [H, L, D]
            a = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1) @ v
            out[dst] = a.transpose(0, 1).reshape(len(src), heads * head_dim).to(out.dtype)
        return out

    def rows_mean(self, src, dst, n_groups, rows_per_group):
        dst.copy_(src.float().reshape(n_groups, rows_per_group, -1).mean(0).to(dst.dtype))
        return dst

    def patchify(self, vid, out):
        T, H, W, C = vid.shape
        x = vid.reshape(T, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 4 * C)
        out.zero_()
        out[:, : 4 * C] = x
        return out

    def unpatchify_euler(self, pred, x_t, out):
        T, H, W, C = out.shape
        p = pred[:, : 4 * C].float().reshape(T, H // 2, W // 2, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(T, H, W, C)
        out.copy_((x_t.float() - p if x_t is not None else p).to(out.dtype))
        return out

    # ------------------------------------------------------------------ VAE side kernels
    def groupnorm_stats(self, x, stats, groups):
        T, H, W, C = x.shape
        xg = _ld(x).double().reshape(T, H * W, groups, C // groups) if x.dtype == H16 else x.double().reshape(T, H * W, groups, C // groups)
        stats[..., 0] = xg.sum(dim=(1, 3))
        stats[..., 1] = xg.pow(2).sum(dim=(1, 3))
        return stats

    def groupnorm_apply(self, x, out, stats, gamma, beta, groups, eps, silu):
        T, H, W, C = x.shape
        n = H * W * (C // groups)
        mean = stats[..., 0] / n
        var = (stats[..., 1] / n - mean * mean).clamp_min(0)
        rstd = 1.0 / torch.sqrt(var + eps)
        mean_c = mean.repeat_interleave(C // groups, dim=1).float()[:, None, None, :]
        rstd_c = rstd.repeat_interleave(C // groups, dim=1).float()[:, None, None, :]
        y = (_ld(x) - mean_c) * rstd_c * gamma + beta
        if silu:
            y = F.silu(y)
        out.copy_(y.to(out.dtype))
        return out

    def im2col_causal(self, x, out, conv):
        g = conv
        kt, kh, kw = g.k
        st, sh, sw = g.stride
        pt, ph, pw = g.pad
        xin = x.reshape(g.T, g.H, g.W, g.Cin)
        if pt > 0:
            head = g.halo[-pt:] if g.halo is not None else xin[:1].expand(pt, g.H, g.W, g.Cin)
            xin = torch.cat([head, xin], dim=0)
        ph_hi = max(0, (g.Ho - 1) * sh + kh - g.H - ph)
        pw_hi = max(0, (g.Wo - 1) * sw + kw - g.W - pw)
        xin = F.pad(xin.permute(0, 3, 1, 2), (pw, pw_hi, ph, ph_hi)).permute(0, 2, 3, 1)
        out.zero_()
        o = out.reshape(g.To, g.Ho, g.Wo, -1)
        tap = 0
        for dt in range(kt):
            for dy in range(kh):
                for dx in range(kw):
                    sl = xin[dt: dt + (g.To - 1) * st + 1: st, dy: dy + (g.Ho - 1) * sh + 1: sh,
                             dx: dx + (g.Wo - 1) * sw + 1: sw]
                    o[..., tap * g.Cin:(tap + 1) * g.Cin] = sl
                    tap += 1
        return out

    def blend_accumulate(self, tile, acc, cnt, wy, wx, y0, x0):
        T, h, w, C = tile.shape
        wgt = wy[:, None] * wx[None, :]
        acc[:, y0:y0 + h, x0:x0 + w] += tile.float() * wgt[None, :, :, None]
        cnt[y0:y0 + h, x0:x0 + w] += wgt

    def blend_finalize(self, acc, cnt, out, scale=1.0, shift=0.0):
        c = out.shape[-1]
        v = acc[..., :c] / cnt.clamp_min(1e-6)[None, :, :, None]
        out.copy_(((v - shift) * scale).to(out.dtype))
        return out

    def affine_slice(self, inp, out, scale=1.0, shift=0.0):
        c = out.shape[-1]
        out.copy_(((inp[..., :c].float() - shift) * scale).to(out.dtype))
        return out
