"""TEST INFRASTRUCTURE -- CPU restatement of the reference NaDiT forward pass (SeedVR2-3B family).

A plain functional torch implementation driven directly by a reference-named state dict,
batch of one clip.  It exists so parity can be re-derived on the GPU box, where
/root/reference is absent; it is itself pinned against the reference implementation
(tests/test_oracle_vs_reference.py when /root/reference is mounted, and the committed
fixtures in tests/golden/ produced by oracle/make_golden.py from the reference).

Reference functions restated (file:line under /root/reference):
  NaDiT.forward                     src/models/dit_3b/nadit.py:190-248
  NaPatchIn / NaPatchOut            src/models/dit_3b/patch/patch_v1.py:76-127
  TimeEmbedding                     src/models/dit_3b/embedding.py:39-62
  get_timestep_embedding            diffusers (restated; embedding.py:50-55 call site)
  NaMMSRTransformerBlock.forward    src/models/dit_3b/nablocks/mmsr_block.py:84-128
  AdaSingle.forward                 src/models/dit_3b/modulation.py:65-118
  CustomRMSNorm.forward             src/models/dit_3b/normalization.py:88-109
  NaSwinAttention.forward           src/models/dit_3b/nablocks/attention/mmattn.py:161-271
  NaMMRotaryEmbedding3d             src/models/dit_3b/rope.py:88-176 (+ rotary_embedding_torch)
  pytorch_varlen_attention          src/models/dit_3b/attention.py:27-64
  SwiGLUMLP.forward                 src/models/dit_3b/mlp.py:46-62
  window partition                  src/models/dit_3b/window.py:28-83 (via the package's windows.py,
                                    which is separately checked against the reference functions)
The 7B family (src/models/dit_7b: nadit.py:39-200, nablocks/mmsr_block.py:33-250, rope.py:28-111, mlp.py:28-43) is the
same graph with: separate vid / txt weights in every block, a full last block, a biased GELU(tanh) MLP, RoPE on the
video tokens only ("pixel" frequencies, positions linspace(-1, 1, n) per window axis, 60 of 128 dims) and no output
norm / modulation -- selected by cfg.mlp_type / rope_type / out_norm / last_vid_only.
The "vid_out_ada" cache-key collision is reproduced: the output
modulation reuses the *attn* slot of the timestep embedding.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F


def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)


def rms_norm(x: torch.Tensor, eps: float, weight=None) -> torch.Tensor:
    y = x / torch.sqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps)
    return y * weight if weight is not None else y


def rope_angles(freqs: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """pos int [L, 3] -> angles [L, 6 * len(freqs)] laid out [axis0 | axis1 | axis2], every
    frequency repeated for the (even, odd) element pair it rotates."""
    parts = []
    for a in range(3):
        ang = pos[:, a].float()[:, None] * freqs.float()[None, :]
        parts.append(ang.repeat_interleave(2, dim=-1))
    return torch.cat(parts, dim=-1)


def apply_rope(x: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
    """x [L, H, D]; rotates the first ang.shape[-1] dims as interleaved pairs, in fp32."""
    rot = ang.shape[-1]
    xf = x.float()
    xr, rest = xf[..., :rot], xf[..., rot:]
    c, s = ang.cos()[:, None, :], ang.sin()[:, None, :]
    x1, x2 = xr[..., 0::2], xr[..., 1::2]
    half = torch.stack((-x2, x1), dim=-1).flatten(-2)
    return torch.cat((xr * c + half * s, rest), dim=-1).to(x.dtype)


def _w(sd, key, dtype):
    return sd[key].to(dtype)


def dit_forward(sd: Dict[str, torch.Tensor], cfg, vid: torch.Tensor, txt: torch.Tensor,
                timestep: float = 1000.0, dtype=torch.float32, windows_mod=None,
                capture: dict = None) -> torch.Tensor:
    """vid [T, H, W, 33] (noise || condition), txt [Lt, 5120] -> prediction [T, H, W, 16]."""
    if windows_mod is None:
        raise ValueError("pass the package's windows module (keeps oracle import-free of the product)")
    d, H, hd = cfg.vid_dim, cfg.heads, cfg.head_dim
    eps = cfg.norm_eps
    T, Hh, Ww, C = vid.shape
    pt, ph, pw = cfg.patch_size
    assert pt == 1 and Hh % ph == 0 and Ww % pw == 0
    t, h, w = T, Hh // ph, Ww // pw
    N = t * h * w
    Lt = txt.shape[0]

    # --- patch in: "(T t)(H h)(W w) c -> T H W (t h w c)"
    x = vid.to(dtype).reshape(t, h, ph, w, pw, C).permute(0, 1, 3, 2, 4, 5).reshape(N, ph * pw * C)
    x = F.linear(x, _w(sd, "vid_in.proj.weight", dtype), _w(sd, "vid_in.proj.bias", dtype))
    y = F.linear(txt.to(dtype), _w(sd, "txt_in.weight", dtype), _w(sd, "txt_in.bias", dtype))

    # --- timestep embedding
    e = timestep_embedding(torch.tensor([float(timestep)])).to(dtype)
    e = F.silu(F.linear(e, _w(sd, "emb_in.proj_in.weight", dtype), _w(sd, "emb_in.proj_in.bias", dtype)))
    e = F.silu(F.linear(e, _w(sd, "emb_in.proj_hid.weight", dtype), _w(sd, "emb_in.proj_hid.bias", dtype)))
    e = F.linear(e, _w(sd, "emb_in.proj_out.weight", dtype), _w(sd, "emb_in.proj_out.bias", dtype))
    emb = e.reshape(d, 2, 3)          # "b (d l g)": l = (attn, mlp), g = (shift, scale, gate)

    def mod_in(hid, branch, li, slot):
        p = f"blocks.{li}.ada.{branch}."
        name = ("attn", "mlp")[slot]
        return hid * (emb[:, slot, 1] + _w(sd, p + name + "_scale", dtype)) \
            + (emb[:, slot, 0] + _w(sd, p + name + "_shift", dtype))

    def mod_out(hid, branch, li, slot):
        p = f"blocks.{li}.ada.{branch}."
        name = ("attn", "mlp")[slot]
        return hid * (emb[:, slot, 2] + _w(sd, p + name + "_gate", dtype))

    scale = 1.0 / math.sqrt(hd)
    for li in range(cfg.num_layers):
        shared = li >= cfg.mm_layers
        last = li == cfg.num_layers - 1 and getattr(cfg, "last_vid_only", True)
        bv, bt = ("all", "all") if shared else ("vid", "txt")
        p = f"blocks.{li}."
        plan = windows_mod.plan_windows((t, h, w), tuple(cfg.window), cfg.window_method(li))

        # ---- attention branch
        xa = mod_in(rms_norm(x, eps), bv, li, 0)
        ya = rms_norm(y, eps)
        if not last:                       # MMModule(vid_only=is_last_layer) on `ada`
            ya = mod_in(ya, bt, li, 0)
        qkv_v = F.linear(xa, _w(sd, p + f"attn.proj_qkv.{bv}.weight", dtype)).reshape(N, 3, H, hd)
        qkv_t = F.linear(ya, _w(sd, p + f"attn.proj_qkv.{bt}.weight", dtype)).reshape(Lt, 3, H, hd)
        qv = rms_norm(qkv_v[:, 0], eps, _w(sd, p + f"attn.norm_q.{bv}.weight", dtype))
        kv = rms_norm(qkv_v[:, 1], eps, _w(sd, p + f"attn.norm_k.{bv}.weight", dtype))
        qt = rms_norm(qkv_t[:, 0], eps, _w(sd, p + f"attn.norm_q.{bt}.weight", dtype))
        kt = rms_norm(qkv_t[:, 1], eps, _w(sd, p + f"attn.norm_k.{bt}.weight", dtype))
        vv, vt = qkv_v[:, 2], qkv_t[:, 2]
        freqs = sd[p + "attn.rope.rope.freqs"].float()
        if getattr(cfg, "rope_type", "mmrope3d") == "rope3d":
            # 7B: get_axial_freqs(f, h, w) of each window, pos = linspace(-1, 1, n); text tokens are not rotated
            ang = torch.zeros(N, 6 * freqs.numel())
            tokl = torch.from_numpy(plan.tok.astype("int64"))
            for wi in range(plan.n_win):
                rows = tokl[plan.cu[wi]:plan.cu[wi + 1]]
                parts = []
                for a in range(3):
                    lin = torch.linspace(-1, 1, steps=int(plan.shapes[wi][a]))
                    idx = torch.from_numpy(plan.pos[rows.numpy(), a].astype("int64"))
                    parts.append((lin[idx][:, None] * freqs[None, :]).repeat_interleave(2, dim=-1))
                ang[rows] = torch.cat(parts, dim=-1)
            qv, kv = apply_rope(qv, ang), apply_rope(kv, ang)
        else:
            pos_v = torch.from_numpy(plan.pos.astype("int64")).clone()
            pos_v[:, 0] += Lt              # video tokens sit after the text on the temporal axis
            jt = torch.arange(Lt)
            pos_t = torch.stack([jt, jt, jt], dim=-1)
            qv, kv = apply_rope(qv, rope_angles(freqs, pos_v)), apply_rope(kv, rope_angles(freqs, pos_v))
            qt, kt = apply_rope(qt, rope_angles(freqs, pos_t)), apply_rope(kt, rope_angles(freqs, pos_t))

        out_v = torch.empty(N, H, hd, dtype=dtype)
        out_t = torch.zeros(Lt, H, hd, dtype=dtype)
        tok = torch.from_numpy(plan.tok.astype("int64"))
        for wi in range(plan.n_win):
            rows = tok[plan.cu[wi]:plan.cu[wi + 1]]
            q = torch.cat([qv[rows], qt]).transpose(0, 1)       # [H, L, hd]
            k = torch.cat([kv[rows], kt]).transpose(0, 1)
            v = torch.cat([vv[rows], vt]).transpose(0, 1)
            a = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1) @ v
            a = a.transpose(0, 1)
            out_v[rows] = a[: rows.numel()]
            out_t += a[rows.numel():]
        out_t /= plan.n_win                                      # na.repeat_concat_idx coalescing
        av = F.linear(out_v.reshape(N, H * hd), _w(sd, p + f"attn.proj_out.{bv}.weight", dtype),
                      _w(sd, p + f"attn.proj_out.{bv}.bias", dtype))
        at = F.linear(out_t.reshape(Lt, H * hd), _w(sd, p + f"attn.proj_out.{bt}.weight", dtype),
                      _w(sd, p + f"attn.proj_out.{bt}.bias", dtype))
        x = x + mod_out(av, bv, li, 0)
        y = y + (at if last else mod_out(at, bt, li, 0))

        # ---- MLP branch
        def mlp(hid, b):
            if getattr(cfg, "mlp_type", "swiglu") == "normal":
                u = F.linear(hid, _w(sd, p + f"mlp.{b}.proj_in.weight", dtype), _w(sd, p + f"mlp.{b}.proj_in.bias", dtype))
                return F.linear(F.gelu(u, approximate="tanh"), _w(sd, p + f"mlp.{b}.proj_out.weight", dtype),
                                _w(sd, p + f"mlp.{b}.proj_out.bias", dtype))
            g = F.linear(hid, _w(sd, p + f"mlp.{b}.proj_in_gate.weight", dtype))
            u = F.linear(hid, _w(sd, p + f"mlp.{b}.proj_in.weight", dtype))
            return F.linear(F.silu(g) * u, _w(sd, p + f"mlp.{b}.proj_out.weight", dtype))

        x = x + mod_out(mlp(mod_in(rms_norm(x, eps), bv, li, 1), bv), bv, li, 1)
        if not last:
            y = y + mod_out(mlp(mod_in(rms_norm(y, eps), bt, li, 1), bt), bt, li, 1)
        if capture is not None:
            capture[f"block{li}.vid"] = x.clone()
            capture[f"block{li}.txt"] = y.clone()

    # --- output head (attn-slot modulation: cache-key collision in the reference, A9)
    if getattr(cfg, "out_norm", True):
        x = rms_norm(x, eps, _w(sd, "vid_out_norm.weight", dtype))
        x = x * (emb[:, 0, 1] + _w(sd, "vid_out_ada.out_scale", dtype)) \
            + (emb[:, 0, 0] + _w(sd, "vid_out_ada.out_shift", dtype))
    x = F.linear(x, _w(sd, "vid_out.proj.weight", dtype), _w(sd, "vid_out.proj.bias", dtype))
    co = cfg.vid_out_channels
    x = x.reshape(t, h, w, ph, pw, co).permute(0, 1, 3, 2, 4, 5).reshape(T, Hh, Ww, co)
    return x
