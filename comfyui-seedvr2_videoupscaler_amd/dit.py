"""NaDiT (SeedVR2-3B family) forward pass on MI355X: host orchestration of the HIP kernels.

Mirrors ``NaDiT.forward(vid, txt, vid_shape, txt_shape, timestep) -> NaDiTOutput(vid_sample)``
(reference: src/models/dit_3b/nadit.py:190-248) for a batch of one clip, which is how the
runner calls it (src/core/infer.py:361-367 with a single-element list, generation_phases.py:718-735).

Design (MI355X-first, not a module-by-module translation):
  * one activation matrix ``hid`` [N + Lt, d]: video tokens first, the Lt text tokens behind them, so
    the 22 shared-weight blocks run ONE GEMM over both streams (the 10 MM blocks run two);
  * no window-ordered copies: q/k-norm + RoPE run in place on the token-ordered QKV buffer with a
    per-token window-local position table, the attention kernel gathers rows through an index vector
    and scatters its output back, text outputs land in a scratch tail and are mean-pooled;
  * AdaLN modulation is per channel at batch 1 -> all (scale, shift, gate) vectors of all layers are
    built once per step (svr_ada_combine) and consumed as fused prologue/epilogue operands:
    RMSNorm+modulate in one pass, gate*x+residual inside the proj_out / mlp-out GEMM epilogues,
    SiLU(gate)*in inside the MLP-in GEMM epilogue (weights interleaved at load);
  * the reference's ``vid_out_ada`` cache-key collision (it modulates with the *attn* slot of the
    timestep embedding) is reproduced deliberately.

The 7B family (src/models/dit_7b) runs through the same engine: ``cfg.mm_layers ==
num_layers`` (separate vid / txt weights everywhere), ``mlp_type "normal"`` (bias + GELU(tanh) fused into the
MLP-in GEMM epilogue, bias + gate + residual into MLP-out), ``rope_type "rope3d"`` (video tokens only; the
fractional window positions linspace(-1, 1, n) become rows of a per-layer cos/sin table, so the same in-place
q/k-norm + RoPE kernel serves both families), ``out_norm False`` and a full last block.
"""
import math
import os
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import windows
from .config import DiTConfig
from .ops import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SILU, EPI_RESID_GATE, EPI_SWIGLU
from .packing import pack_matrix, pack_swiglu, pack_vec

BF16 = torch.bfloat16
DEFAULT_HID_STORE = "h16"           # storage of the residual stream unless the caller (or SVR_DIT_STREAM) says otherwise: see NaDiTEngine


@dataclass
class NaDiTOutput:
    vid_sample: torch.Tensor


@dataclass
class _Lin:
    w: torch.Tensor
    b: Optional[torch.Tensor]
    n: int
    k: int
    frag: Optional[torch.Tensor] = None     # fragment-ordered copy of w for the persistent GEMM kernel (ops.pack_gemm_frag), or None


def timestep_sinusoid(t: float, dim: int = 256) -> torch.Tensor:
    """diffusers.get_timestep_embedding(flip_sin_to_cos=False, downscale_freq_shift=0), fp32
    (embedding.py:50-55).  Host constant: the one-step sampler always asks for t = T = 1000."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = torch.tensor([float(t)], dtype=torch.float32)[:, None] * freqs[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)


class NaDiTEngine:
    def __init__(self, cfg: DiTConfig, state_dict: Dict[str, torch.Tensor], ops, hid_fp32: Optional[bool] = None,
                 hid_store: Optional[str] = None, overflow_guard: bool = True):
        """``hid_store`` ("fp32" | "h16" | "bf16"): how the residual stream ``hid`` is held.  Its only readers are RMSNorm and the
        gate + residual GEMM epilogues -- every MFMA operand stays bf16 -- so it can be WIDE without touching the matrix pipe:
        bf16 costs 64 roundings of the stream per 32-layer pass (49.1 dB end to end at production width, tools/error_budget.py: the
        most expensive store of the whole chain), fp32 none (rounds 3-4), "h16" (round 5; an IEEE half of x * 2^-6, ops.H16: 11
        significant bits in two bytes, range +-4.2e6) -0.07 dB against fp32 at half the bytes: the attn-out / mlp-out epilogues
        read and write 2 instead of 4 B per element and RMSNorm reads 2.  Default on the bf16 backends: "h16" (measured on MI355X at
        BASELINE config 3, same box: DiT step 1 322 -> 1 289 ms; production-width chain 50.9 dB either way, 36-layer 7B 4.7e-3
        against 4.4e-3: profiles/r5_dit_stream_ab.txt).
        ``hid_fp32`` (rounds 3-4 spelling): True -> "fp32", False -> "bf16".
        ``overflow_guard``: h16 ends at +-4.2e6 where fp32 / bf16 go on; forward() looks at one sum of its result (one host sync per
        call) and, when it is not finite under an h16 stream although the inputs were, repeats the call with an fp32 stream and warns."""
        self.cfg, self.ops = cfg, ops
        if hid_store is None and hid_fp32 is None:
            hid_store = os.environ.get("SVR_DIT_STREAM") or DEFAULT_HID_STORE      # (the env var: A/B runs of the parity tests)
        elif hid_store is None:
            hid_store = "fp32" if hid_fp32 else "bf16"
        if hid_store not in ("fp32", "h16", "bf16"):
            raise ValueError(f"hid_store must be 'fp32', 'h16' or 'bf16', got {hid_store!r}")
        if getattr(ops, "act_dtype", BF16) != BF16:      # (the exact-arithmetic CPU double keeps everything in its one dtype)
            hid_store = "fp32" if ops.act_dtype == torch.float32 else hid_store
        self.hid_store = hid_store
        self.hid_dtype = {"fp32": torch.float32, "h16": torch.float16, "bf16": None}[hid_store]     # None: the ops' activation dtype
        self.overflow_guard, self.overflow_reruns = bool(overflow_guard), 0
        dev = ops.device
        self.device = dev
        d, inner = cfg.vid_dim, cfg.heads * cfg.head_dim
        if cfg.head_dim != 128:
            raise ValueError("the window-attention kernel is built for head_dim 128")
        sd = state_dict

        def frag(wp):     # (block weights only: the GEMMs that see every video token)
            return ops.pack_gemm_frag(wp) if hasattr(ops, "pack_gemm_frag") else None

        def lin(name, bias=True, big=False):
            w = sd[name + ".weight"]
            wp, kpad = pack_matrix(w, dev), (w.shape[1] + 63) // 64 * 64
            return _Lin(wp, pack_vec(sd[name + ".bias"], dev) if bias else None, w.shape[0], kpad, frag(wp) if big else None)

        self.vid_in = lin("vid_in.proj")
        self.txt_in = lin("txt_in")
        self.emb_in = [lin("emb_in.proj_in"), lin("emb_in.proj_hid"), lin("emb_in.proj_out")]
        self.vid_out = lin("vid_out.proj")
        self.out_norm_w = pack_vec(sd["vid_out_norm.weight"], dev) if cfg.out_norm else None
        self.kpad_in = self.vid_in.k

        # ---- per-block weights
        self.blocks = []
        ada_rows, ada_slots = [], []

        def add_ada(key, slot):
            ada_rows.append(sd[key].to(BF16))
            ada_slots.append(slot)
            return len(ada_rows) - 1

        for i in range(cfg.num_layers):
            shared = i >= cfg.mm_layers
            p = f"blocks.{i}."
            blk = {"shared": shared}
            for stream, b in (("vid", "all" if shared else "vid"), ("txt", "all" if shared else "txt")):
                if shared and stream == "txt":
                    blk["txt"] = blk["vid"]
                    continue
                s = {}
                big = stream == "vid"             # (the text stream's 58 rows per window never reach the persistent kernel)
                s["qkv"] = lin(p + f"attn.proj_qkv.{b}", bias=False, big=big)
                s["out"] = lin(p + f"attn.proj_out.{b}", big=big)
                s["wq"] = pack_vec(sd[p + f"attn.norm_q.{b}.weight"], dev)
                s["wk"] = pack_vec(sd[p + f"attn.norm_k.{b}.weight"], dev)
                if cfg.mlp_type == "normal":
                    s["mlp_in"] = lin(p + f"mlp.{b}.proj_in", big=big)
                    s["mlp_out"] = lin(p + f"mlp.{b}.proj_out", big=big)
                else:
                    wg, wi = sd[p + f"mlp.{b}.proj_in_gate.weight"], sd[p + f"mlp.{b}.proj_in.weight"]
                    wsw = pack_swiglu(wg, wi, dev)
                    s["mlp_in"] = _Lin(wsw, None, 2 * wg.shape[0], wg.shape[1], frag(wsw) if big else None)
                    s["mlp_out"] = lin(p + f"mlp.{b}.proj_out", bias=False, big=big)
                # AdaSingle parameters; slot = l*3 + g with l in (attn, mlp), g in (shift, scale, gate)
                s["ada"] = {}
                for l, lname in enumerate(("attn", "mlp")):
                    for gi, gname in enumerate(("shift", "scale", "gate")):
                        s["ada"][(lname, gname)] = add_ada(p + f"ada.{b}.{lname}_{gname}", l * 3 + gi)
                blk[stream] = s
            blk["freqs"] = sd[p + "attn.rope.rope.freqs"].float().cpu()
            self.blocks.append(blk)
        if cfg.out_norm:   # output modulation reuses the attn slot (l = 0) of the embedding: shift g=0, scale g=1
            self.ada_out_shift = add_ada("vid_out_ada.out_shift", 0)
            self.ada_out_scale = add_ada("vid_out_ada.out_scale", 1)
        self.ada_params = torch.stack(ada_rows).to(dev).contiguous()
        self.ada_slots = torch.tensor(ada_slots, dtype=torch.int32, device=dev)
        self._plan_cache = {}
        self._rope_cache = {}
        self.tap = None     # tests only: callable(tag, tensor) shown the residual stream in front of every block and behind the last

    # ------------------------------------------------------------------ nn.Module-shaped probes
    # The reference's phase code asks its models where and what they are the nn.Module way -- next(model.parameters()).device
    # / .dtype (generation_phases.py:298, 620, 708-712), .eval(), .to(device), .requires_grad_(False): answered here so that
    # code can drive the engines unchanged.  Weights are resident, pre-tiled tensors; moving them is not supported.
    def parameters(self):
        seen = set()

        def walk(o):
            if torch.is_tensor(o):
                if o.is_floating_point() and id(o) not in seen:
                    seen.add(id(o))
                    yield o
            elif isinstance(o, dict):
                for v in o.values():
                    yield from walk(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    yield from walk(v)
            elif hasattr(o, "__dataclass_fields__"):
                for f in o.__dataclass_fields__:
                    yield from walk(getattr(o, f))
        for name, v in vars(self).items():
            if name not in ("ops", "cfg") and not name.startswith("_"):
                yield from walk(v)

    def eval(self):
        return self

    def requires_grad_(self, flag: bool = False):
        return self

    def to(self, *args, **kwargs):
        dev = next((a for a in args if isinstance(a, (str, torch.device))), kwargs.get("device"))
        if dev is not None and torch.device(dev).type != self.device.type:
            raise NotImplementedError(f"{type(self).__name__} weights are resident on {self.device} (pre-tiled for the MFMA kernels); "
                                      "build a new engine on the target device instead of moving this one")
        return self

    @property
    def dtype(self):
        return self.ops.act_dtype

    # ------------------------------------------------------------------ host-side index plans
    def _plan(self, size, method, Lt):
        key = (size, method, Lt)
        if key in self._plan_cache:
            return self._plan_cache[key]
        plan = windows.plan_windows(size, tuple(self.cfg.window), method)
        N = size[0] * size[1] * size[2]
        n_win = plan.n_win
        lens = np.diff(plan.cu)
        total = N + n_win * Lt
        seq_rows = np.empty(total, dtype=np.int32)
        out_rows = np.empty(total, dtype=np.int32)
        cu = np.zeros(n_win + 1, dtype=np.int32)
        txt_src = N + np.arange(Lt, dtype=np.int32)
        o = 0
        for w in range(n_win):
            rows = plan.tok[plan.cu[w]:plan.cu[w + 1]]
            seq_rows[o:o + lens[w]] = rows
            out_rows[o:o + lens[w]] = rows
            o += lens[w]
            seq_rows[o:o + Lt] = txt_src                       # every window sees the same text rows
            out_rows[o:o + Lt] = N + Lt + w * Lt + np.arange(Lt)   # per-window text outputs -> scratch tail
            o += Lt
            cu[w + 1] = o
        dev = self.device
        res = dict(n_win=n_win, max_len=int(lens.max()) + Lt,
                   seq_rows=torch.from_numpy(seq_rows).to(dev), out_rows=torch.from_numpy(out_rows).to(dev),
                   cu=torch.from_numpy(cu).to(dev), pos=torch.from_numpy(plan.pos.copy()).to(dev))
        if self.cfg.rope_type == "rope3d":
            # 7B: position along an axis of extent n is linspace(-1, 1, n)[idx]; one table row per (n, idx), row 0 = angle 0
            extents = sorted({int(v) for v in np.unique(plan.shapes)})
            offs, values = {}, [0.0]
            for n in extents:
                offs[n] = len(values)
                values.extend(torch.linspace(-1, 1, steps=n).tolist())
            rows = np.zeros((N, 3), dtype=np.int16)
            for wi in range(n_win):
                tok = plan.tok[plan.cu[wi]:plan.cu[wi + 1]]
                for a in range(3):
                    rows[tok, a] = offs[int(plan.shapes[wi][a])] + plan.pos[tok, a]
            res["pos"] = torch.from_numpy(rows).to(dev)
            res["pos_values"] = torch.tensor(values, dtype=torch.float32)
        self._plan_cache[key] = res
        return res

    def _rope3d_tables(self, freqs: torch.Tensor, values: torch.Tensor):
        key = (tuple(freqs.tolist()), tuple(values.tolist()))
        if key not in self._rope_cache:
            ang = values[:, None] * freqs[None, :]                                      # fp32, as the reference
            self._rope_cache[key] = (ang.cos().to(self.device).contiguous(), ang.sin().to(self.device).contiguous())
        return self._rope_cache[key]

    def _rope_tables(self, freqs: torch.Tensor, n_pos: int):
        key = (tuple(freqs.tolist()), n_pos)
        if key not in self._rope_cache:
            ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * freqs[None, :]   # fp32, as the reference
            self._rope_cache[key] = (ang.cos().to(self.device).contiguous(), ang.sin().to(self.device).contiguous())
        return self._rope_cache[key]

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, vid: torch.Tensor, txt: torch.Tensor, timestep: float = 1000.0,
                x_t: Optional[torch.Tensor] = None) -> torch.Tensor:
        """vid [T, H, W, 33] bf16 (x_t || condition), txt [Lt, txt_in_dim] bf16.
        Returns the model prediction [T, H, W, 16]; with ``x_t`` given returns the one-step Euler
        endpoint x_t - pred instead (fused into the un-patchify kernel)."""
        out = self._forward(vid, txt, timestep, x_t, self.hid_dtype)
        # the guard is an h16 affair: an fp32 / bf16 stream returns without looking (no reduction, no host sync)
        if self.hid_store != "h16" or not self.overflow_guard or bool(torch.isfinite(out.sum(dtype=torch.float32))):
            return out
        import warnings
        if not (bool(torch.isfinite(vid.float().sum())) and bool(torch.isfinite(txt.float().sum()))
                and (x_t is None or bool(torch.isfinite(x_t.float().sum())))):
            warnings.warn("NaDiTEngine: the input of this call is not finite; its output is returned as computed (not an h16 range "
                          "problem, no fp32 re-run)", RuntimeWarning, stacklevel=2)
            return out
        warnings.warn("NaDiTEngine: non-finite output with an h16 residual stream (an activation beyond +-4.2e6?); repeating the call "
                      "with an fp32 stream", RuntimeWarning, stacklevel=2)
        self.overflow_reruns += 1
        # (the store kind travels as an argument: the engine's own state is not touched, so a concurrent or re-entrant call on the
        # same engine keeps its h16 stream)
        return self._forward(vid, txt, timestep, x_t, torch.float32)

    def _forward(self, vid, txt, timestep, x_t, hid_dtype):
        cfg, ops = self.cfg, self.ops
        d, heads, hd = cfg.vid_dim, cfg.heads, cfg.head_dim
        inner = heads * hd
        T, H, W, Cin = vid.shape
        assert Cin == cfg.vid_in_channels and H % 2 == 0 and W % 2 == 0
        t, h, w = T, H // 2, W // 2
        N, Lt = t * h * w, txt.shape[0]
        R = N + Lt
        eps = cfg.norm_eps

        hid = ops.empty(R, d, dtype=hid_dtype)
        hf = hid.dtype in (torch.float32, torch.float16)     # a WIDE stream (fp32 or h16): the ops derive the kind from the dtype
        a0 = ops.empty(N, self.kpad_in)
        ops.patchify(vid.contiguous(), a0)
        ops.gemm(a0, self.vid_in.w, hid[:N], N=d, K=self.kpad_in, bias=self.vid_in.b, out_f32=hf)
        del a0
        ops.gemm(txt.contiguous(), self.txt_in.w, hid[N:], N=d, K=self.txt_in.k, bias=self.txt_in.b, out_f32=hf)

        # ---- timestep embedding -> all AdaLN vectors of the step
        e = timestep_sinusoid(timestep).to(device=self.device, dtype=ops.act_dtype)
        e1, e2 = ops.empty(1, d), ops.empty(1, d)
        emb = ops.empty(1, cfg.emb_dim)
        ops.gemm(e, self.emb_in[0].w, e1, N=d, K=256, bias=self.emb_in[0].b, epilogue=EPI_BIAS_SILU)
        ops.gemm(e1, self.emb_in[1].w, e2, N=d, K=d, bias=self.emb_in[1].b, epilogue=EPI_BIAS_SILU)
        ops.gemm(e2, self.emb_in[2].w, emb, N=cfg.emb_dim, K=d, bias=self.emb_in[2].b)
        mod = ops.empty(self.ada_params.shape[0], d, dtype=torch.float32)
        ops.ada_combine(emb.reshape(-1), self.ada_params, self.ada_slots, mod)

        xn = ops.empty(R, d)
        qkv = ops.empty(R, 3 * inner)
        h1 = ops.empty(R, cfg.mlp_hidden)
        jt = torch.arange(Lt, dtype=torch.int16)
        pos_t = torch.stack([jt, jt, jt], dim=-1).contiguous().to(self.device)
        scale = 1.0 / math.sqrt(hd)

        rope3d = cfg.rope_type == "rope3d"
        if rope3d:
            pos_t = torch.zeros(Lt, 3, dtype=torch.int16, device=self.device)           # table row 0: no rotation
        for li, blk in enumerate(self.blocks):
            if self.tap is not None:
                self.tap(f"hid{li}", hid)
            final = li == cfg.num_layers - 1          # after this block's attention the text stream is dead
            last = final and cfg.last_vid_only        # 3B: the last block's text branch is not modulated
            shared = blk["shared"]
            sv, st = blk["vid"], blk["txt"]
            plan = self._plan((t, h, w), cfg.window_method(li), Lt)
            n_win = plan["n_win"]
            if rope3d:
                cos_t, sin_t = self._rope3d_tables(blk["freqs"], plan["pos_values"])
            else:
                cos_t, sin_t = self._rope_tables(blk["freqs"], max(Lt + t, h, w) + 16)

            # ---- attention branch
            ops.rmsnorm_mod(hid[:N], xn[:N], eps, scale=mod[sv["ada"][("attn", "scale")]],
                            shift=mod[sv["ada"][("attn", "shift")]])
            if last:      # MMModule(ada, vid_only=True): the text stream is normalised but not modulated
                ops.rmsnorm_mod(hid[N:], xn[N:], eps)
            else:
                ops.rmsnorm_mod(hid[N:], xn[N:], eps, scale=mod[st["ada"][("attn", "scale")]],
                                shift=mod[st["ada"][("attn", "shift")]])
            if shared:
                ops.gemm(xn, sv["qkv"].w, qkv, N=3 * inner, K=d, W_frag=sv["qkv"].frag)
            else:
                ops.gemm(xn[:N], sv["qkv"].w, qkv[:N], N=3 * inner, K=d, W_frag=sv["qkv"].frag)
                ops.gemm(xn[N:], st["qkv"].w, qkv[N:], N=3 * inner, K=d)
            ops.qknorm_rope(qkv[:N], heads, plan["pos"], 0 if rope3d else Lt, cos_t, sin_t, sv["wq"], sv["wk"], eps)
            ops.qknorm_rope(qkv[N:], heads, pos_t, 0, cos_t, sin_t, st["wq"], st["wk"], eps)
            att = ops.empty(R + n_win * Lt, inner)
            ops.attn_varlen(qkv, att, plan["seq_rows"], plan["out_rows"], plan["cu"], plan["max_len"], heads, hd, scale)
            ops.rows_mean(att[R:], att[N:R], n_win, Lt)
            g_v = mod[sv["ada"][("attn", "gate")]]
            if shared and not final:
                ops.gemm(att[:R], sv["out"].w, hid, N=d, K=inner, bias=sv["out"].b, epilogue=EPI_RESID_GATE,
                         gate=g_v, resid=hid, out_f32=hf, W_frag=sv["out"].frag)
            else:
                ops.gemm(att[:N], sv["out"].w, hid[:N], N=d, K=inner, bias=sv["out"].b, epilogue=EPI_RESID_GATE,
                         gate=g_v, resid=hid[:N], out_f32=hf, W_frag=sv["out"].frag)
                if not final:  # the text stream is dead after the last block's attention
                    ops.gemm(att[N:R], st["out"].w, hid[N:], N=d, K=inner, bias=st["out"].b, out_f32=hf,
                             epilogue=EPI_RESID_GATE, gate=mod[st["ada"][("attn", "gate")]], resid=hid[N:])
            del att

            # ---- MLP branch (3B: SwiGLU, 7B: GELU); video only in the last block (its text output is never read)
            hm = cfg.mlp_hidden
            ops.rmsnorm_mod(hid[:N], xn[:N], eps, scale=mod[sv["ada"][("mlp", "scale")]],
                            shift=mod[sv["ada"][("mlp", "shift")]])
            if not final:
                ops.rmsnorm_mod(hid[N:], xn[N:], eps, scale=mod[st["ada"][("mlp", "scale")]],
                                shift=mod[st["ada"][("mlp", "shift")]])
            gm_v = mod[sv["ada"][("mlp", "gate")]]

            def mlp(rows, s_, gate, dst=None):
                """``dst``: where the branch's output (residual added) goes instead of back into ``hid``."""
                if cfg.mlp_type == "normal":
                    ops.gemm(xn[rows], s_["mlp_in"].w, h1[rows], N=hm, K=d, bias=s_["mlp_in"].b, epilogue=EPI_BIAS_GELU,
                             W_frag=s_["mlp_in"].frag)
                else:
                    ops.gemm(xn[rows], s_["mlp_in"].w, h1[rows], N=2 * hm, K=d, epilogue=EPI_SWIGLU, W_frag=s_["mlp_in"].frag)
                o = hid[rows] if dst is None else dst
                ops.gemm(h1[rows], s_["mlp_out"].w, o, N=d, K=hm, bias=s_["mlp_out"].b, out_f32=o.dtype in (torch.float32, torch.float16),
                         epilogue=EPI_RESID_GATE, gate=gate, resid=hid[rows], W_frag=s_["mlp_out"].frag)

            if shared and not final:
                mlp(slice(0, R), sv, gm_v)
            else:
                # (no output norm -- the 7B family: vid_out reads the stream as an MFMA operand, so the last block leaves it in
                # the activation dtype)
                mlp(slice(0, N), sv, gm_v, dst=xn[:N] if final and not cfg.out_norm and hf else None)
                if not final:
                    mlp(slice(N, R), st, mod[st["ada"][("mlp", "gate")]])

        # ---- output head
        if self.tap is not None:
            self.tap(f"hid{cfg.num_layers}", xn if (not cfg.out_norm and hf) else hid)
        pred = ops.empty(N, cfg.patch_out_dim)
        if cfg.out_norm:
            ops.rmsnorm_mod(hid[:N], xn[:N], eps, w=self.out_norm_w, scale=mod[self.ada_out_scale],
                            shift=mod[self.ada_out_shift])
            ops.gemm(xn[:N], self.vid_out.w, pred, N=cfg.patch_out_dim, K=d, bias=self.vid_out.b)
        else:
            ops.gemm(xn[:N] if hf else hid[:N], self.vid_out.w, pred, N=cfg.patch_out_dim, K=d, bias=self.vid_out.b)
        out = ops.empty(T, H, W, cfg.vid_out_channels)
        ops.unpatchify_euler(pred, None if x_t is None else x_t.contiguous(), out)
        return out

    # reference-shaped call: NaDiT.forward(vid (L, 33), txt (l, 5120), vid_shape (B, 3), txt_shape (B, 1), timestep (B,)) -- a batch is the
    # clips' rows concatenated (na.flatten, infer.py:355-357); clips never interact, so each runs on its own and the rows are re-joined
    def __call__(self, vid, txt, vid_shape, txt_shape=None, timestep=1000.0):
        B = int(vid_shape.shape[0])
        if txt_shape is None:
            if B != 1:
                raise ValueError("txt_shape is required for a batch of clips")
            txt_lens = [int(txt.shape[0])]
        else:
            txt_lens = [int(v) for v in txt_shape.reshape(B, -1).prod(dim=1).tolist()]
        if torch.is_tensor(timestep):
            tl = [float(v) for v in timestep.reshape(-1).tolist()]
            tl = tl * B if len(tl) == 1 else tl
        else:
            tl = [float(timestep)] * B
        if len(tl) != B or sum(txt_lens) != txt.shape[0]:
            raise ValueError("timestep / txt_shape do not match the batch")
        outs, v0, t0 = [], 0, 0
        for b in range(B):
            T, H, W = (int(v) for v in vid_shape[b].tolist())
            n = T * H * W
            out = self.forward(vid[v0:v0 + n].reshape(T, H, W, -1).to(self.ops.act_dtype), txt[t0:t0 + txt_lens[b]].to(self.ops.act_dtype), tl[b])
            outs.append(out.reshape(n, -1))
            v0, t0 = v0 + n, t0 + txt_lens[b]
        if v0 != vid.shape[0]:
            raise ValueError("vid_shape does not cover the rows of vid")
        return NaDiTOutput(vid_sample=outs[0] if B == 1 else torch.cat(outs, dim=0))
