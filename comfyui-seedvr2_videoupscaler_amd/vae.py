"""Causal-Conv3d video VAE (SeedVR2 "video_vae_v3", s8 / t4 / c16) on MI355X: host orchestration.

Mirrors ``VideoAutoencoderKLWrapper.encode / .decode(x, tiled, tile_size, tile_overlap)``
(reference: src/models/video_vae_v3/modules/attn_video_vae.py:1680-1699) and the runner-level
latent scaling of ``VideoDiffusionInfer.vae_encode / vae_decode`` (src/core/infer.py:117-278).

Design (MI355X-first):
  * activations are NDHWC bf16 ([T, H, W, C]); every convolution is one implicit-GEMM launch that
    gathers its taps straight from the input tensor -- no im2col, no padded copies, no NCDHW;
  * the causal temporal head is an index clamp (first slice) or a 1-2 frame halo tensor carried per
    layer between temporal slices (the reference's per-conv ``memory``, causal_inflation_lib.py:260-278);
    slices are sized from an activation-byte budget instead of the reference's fixed 4 frames (the
    result is independent of the slice size), so a 1024-px tile runs as ONE slice;
  * GroupNorm statistics are per frame (causal_norm_wrapper), fp64-reduced; apply+SiLU is one pass;
  * residual adds, biases and the 3D pixel-shuffle of Upsample3D are GEMM epilogues;
  * spatial tiling reproduces tiled_encode / tiled_decode tile for tile (per-tile GroupNorm and
    attention change results, so this is semantics, not an optimisation); blending accumulates fp32.
"""
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from .config import VAEConfig
from . import subpixel
from .ops import EPI_BIAS, EPI_RESID_GATE, Conv3dGeom, PhaseScatter, PixelShuffleGeom
from .packing import BF16, pack_conv3d, pack_matrix, pack_vec

BF16 = torch.bfloat16


@dataclass
class _Conv:
    name: str
    w: torch.Tensor
    b: torch.Tensor
    cin: int            # channels the kernel sees (after padding thin inputs)
    cout: int
    k: Tuple[int, int, int]
    stride: Tuple[int, int, int] = (1, 1, 1)
    pad_lo: int = 1     # spatial zero pad, low side
    pad_hi: int = 1     # spatial zero pad, high side
    thin: bool = False  # Cin < 64: im2col + plain GEMM
    w_frag: Optional[torch.Tensor] = None   # MFMA-fragment-ordered copy of w for the LDS-halo conv kernel (ops.pack_conv_frag)
    # causal head (kt = 3, stride 1): the same conv for output frame 0 of a clip, whose three temporal taps all fall on the
    # replicated first frame (causal_inflation_lib.py:422-437) -> kt = 2 conv with (hi, lo) = W0+W1+W2 split into two bf16 terms
    head: Optional["_Conv"] = None


@dataclass
class _Norm:
    gamma: torch.Tensor
    beta: torch.Tensor


@dataclass
class _Resnet:
    norm1: _Norm
    conv1: _Conv
    norm2: _Norm
    conv2: _Conv
    shortcut: Optional[_Conv]


@dataclass
class _Attn:
    norm: _Norm
    qkv_w: torch.Tensor
    qkv_b: torch.Tensor
    out_w: torch.Tensor
    out_b: torch.Tensor
    dim: int


@dataclass
class _Up:
    upscale_w: torch.Tensor
    upscale_b: torch.Tensor
    conv: _Conv
    temporal: bool
    C: int
    # sub-pixel form (subpixel.py): signature -> (temporal window length, [(py, px, packed weight [Cout, kt'*2*2*C], bias [Cout],
    # bias_border [3, Cout], fragment-ordered weight copy or None)]) for the tap patterns this upsampler meets (1 if
    # spatial-only, 5 if temporal)
    merged: Optional[dict] = None


def _tile_ranges(total: int, tile: int, overlap: int) -> List[Tuple[int, int]]:
    """Tile starts of tiled_encode / tiled_decode (attn_video_vae.py:1363-1372): stride = tile - overlap,
    tiles that lie entirely inside the previous tile's overlap are skipped."""
    stride = max(1, tile - overlap)
    out = []
    for s in range(0, total, stride):
        e = min(s + tile, total)
        if s > 0 and (e - s) <= overlap:
            continue
        out.append((s, e))
    return out


def _edge_weights(length: int, ov: int, fade_lo: bool, fade_hi: bool, ramp: Optional[torch.Tensor]) -> torch.Tensor:
    w = torch.ones(length, dtype=torch.float32)
    ov = max(0, min(ov, length - 1))
    if ov > 0:
        if fade_lo:
            w[:ov] = ramp[:ov]
        if fade_hi:
            w[-ov:] = 1 - ramp[:ov]
    return w


def _cos_ramp(n: int) -> Optional[torch.Tensor]:
    if n <= 0:
        return None
    return 0.5 - 0.5 * torch.cos(torch.linspace(0, 1, steps=n, dtype=torch.float32) * math.pi)


class VideoVAEEngine:
    def __init__(self, cfg: VAEConfig, state_dict: Dict[str, torch.Tensor], ops,
                 act_budget_bytes: Optional[int] = None, merge_upsamplers: bool = True, merge_causal_head: bool = True,
                 trunk_fp32: Optional[bool] = None, branch_fp32: Optional[bool] = None, tile_streams: int = 1,
                 trunk_store: Optional[str] = None, branch_store: Optional[str] = None, overflow_guard: bool = True):
        """``act_budget_bytes``: bound on the widest activation of one temporal slice, from which encode_clip / decode_clip derive the
        slice length (results do not depend on it, bit for bit).  An explicit argument wins; ``None`` (default) takes env
        SVR_VAE_ACT_BUDGET_GIB (A/B runs; validated) or else 24 GiB, clamped to 1/8 of the device's free memory divided by
        ``tile_streams`` (the peak of a slice is several times its widest activation; a smaller HBM partition must not OOM).
        24 GiB = the 9 latents (33 frames) of a 1024-px decoder tile in ONE slice: no carried-frame copies between slices, no second
        set of launches; same box, BASELINE config 3: decode 5 146 / 5 172 ms at 12 GiB (two slices) -> 5 129 / 5 153 ms
        (profiles/r5_vae_slice_budget_ab.txt); 48 GiB changes nothing further.  Peak memory of the config-3 step stays far below 288 GB.
        ``tile_streams`` (default 1): spatial tiles of a tiled encode / decode are independent until the blend
        (attn_video_vae.py:1302-1630), so they CAN be issued round-robin onto several HIP streams, the HBM-bound passes of one tile
        (GroupNorm apply, statistics, layout copies) in the shadow of another tile's MFMA-bound convolutions; the blends are issued
        on the caller's stream in tile order, so the result is bit-identical to one stream (_run_tiles; tested).  MEASURED on
        MI355X at BASELINE config 3, round 4, same box: 9.89 s per step on one stream, 9.85 s on two, 9.87 s on three -- the launches
        time-slice (every kernel's duration doubles) instead of overlapping: the chip is power-limited, idle issue slots of the conv
        kernel are not free capacity (profiles/r4_tile_streams_ab.txt).  Kept as an option, off by default.
        ``merge_upsamplers``: run the spatial-only upsampler (upscale_conv + pixel shuffle + 3x3x3 conv) as four sub-pixel
        convs over its low-resolution input (subpixel.py) -- same function, 12 instead of 28 MACs per output voxel and channel
        pair, no upsampled intermediate; False keeps the reference's two steps.
        ``merge_causal_head``: output frame 0 of a clip sees the replicated first frame under all three temporal taps
        (extend_head, causal_inflation_lib.py:422-437), so it is computed as (W0+W1+W2) * x[0] with the sum held as two bf16
        terms (hi + lo, exact to 2^-17): 18 instead of 27 MACs per voxel and channel pair, same result.  (Rounding the sum
        to ONE bf16 term -- and merging frame 1's two replicated taps the same way -- would save three times as much but costs
        0.4-0.8 dB against the fp32 reference: measured, not shipped.)  False keeps three taps on every frame.
        ``trunk_store`` / ``branch_store`` ("bf16" | "h16" | "fp32"; default "h16" for both on a bf16 backend): how the tensors
        that are NOT MFMA operands are held.  The residual TRUNK (ResnetBlock3D / attention / shortcut-conv outputs,
        attn_video_vae.py:311-362, 615-665) is read only by GroupNorm and the next residual add; it is stored narrow (bf16) exactly
        where a conv reads it directly as an MFMA operand (in front of a down/upsampler or a shortcut conv): 5 instead of 17 bf16
        roundings on the decoder's skip path.  The BRANCH tensor is conv1's output inside a block, read only by norm2.
        "h16" (round 4, ABI v6; ops.H16): an IEEE half holding x * 2^-6 -- 11 significant bits in bf16's two bytes, range +-4.2e6.
        On the production-width chain (tests/golden/pipeline_prod.pt, CPU double of the C ABI, tools/error_budget.py): trunk / branch
        fp32 / fp32 50.01 dB, h16 / h16 49.99 dB, fp32 / bf16 (round 3's default) 49.76 dB, bf16 / bf16 (round 2) 48.20 dB -- h16 is
        as good as fp32 at half the bytes: GroupNorm-apply reads 2 instead of 4 B per element, conv2 epilogues store half as much
        (round 3's fp32 trunk cost +2.4 % of the BASELINE config 3 step, its fp32 branch another +2.1 %).
        ``trunk_fp32`` / ``branch_fp32`` (rounds 2-3, kept for A/B): True -> "fp32", False -> "bf16" for the respective tensor.
        ``overflow_guard`` (default on): h16 ends at +-4.2e6 where bf16 / fp32 (what the reference runs in) go on to 3e38.  The
        synthetic and fixture weights stay orders of magnitude below that, but an un-normalised activation of a real checkpoint
        beyond it would turn into inf on store and into NaN behind the next GroupNorm.  encode() / decode() therefore look at ONE
        fp32 sum of their result (a single reduction over the output, one host sync per call) and, when it is not finite while an
        h16 store is in use, run the call again with those stores in fp32 -- same kernels, their fp32 instances -- and warn."""
        self.cfg, self.ops = cfg, ops
        kinds = {"bf16": None, "h16": torch.float16, "fp32": torch.float32}      # None: the ops' activation dtype
        # (the exact-arithmetic CPU double of the C ABI -- act_dtype fp32, host-logic tests -- keeps everything in its one dtype)
        default = "h16" if getattr(ops, "act_dtype", BF16) == BF16 else "bf16"
        if trunk_store is None:
            trunk_store = default if trunk_fp32 is None else ("fp32" if trunk_fp32 else "bf16")
        if branch_store is None:
            branch_store = (default if trunk_store != "bf16" else "bf16") if branch_fp32 is None else \
                ("fp32" if (branch_fp32 and trunk_store != "bf16") else "bf16")
        self.trunk_store, self.branch_store = trunk_store, branch_store
        self.trunk_dtype, self.branch_dtype = kinds[trunk_store], kinds[branch_store]
        self.overflow_guard = bool(overflow_guard)
        self.overflow_reruns = 0            # calls the guard had to repeat with fp32 stores
        self.device = ops.device
        self.tile_streams = int(tile_streams)
        self.act_budget_bytes = self._resolve_act_budget(act_budget_bytes, ops.device, self.tile_streams)
        self.sample_dtype = None            # tools/error_budget.py only (CPU double): dtype of the decoder's own output tile; the HIP
                                            # blend kernels take bf16 tiles, so the product leaves it at None (measured: +0.1 dB for fp32)
        self._streams = []
        self._edges = {}
        sd, dev = state_dict, ops.device
        ch = cfg.block_out_channels
        n = len(ch)

        def conv(name, stride=(1, 1, 1), pad=(1, 1), cin_pad=None, w=None):
            w = sd[name + ".weight"] if w is None else w
            co, ci, kt, kh, kw = w.shape
            thin = ci < 64
            if thin:
                cin_pad = cin_pad or (ci + 3) // 4 * 4
            wp = pack_conv3d(w, dev, cin_pad)
            frag = None
            if (kh, kw) == (3, 3) and tuple(stride) == (1, 1, 1) and tuple(pad) == (1, 1) and not thin \
                    and hasattr(ops, "pack_conv_frag"):
                frag = ops.pack_conv_frag(wp, kt, ci, co)
            head = None
            if merge_causal_head and kt == 3 and tuple(stride) == (1, 1, 1) and not thin:
                # (weights as the reference holds them: cast to bf16 at load, model_loader.py:583-584; summed in fp32)
                wsum = w.to(device=dev, dtype=BF16).float().sum(2)
                hi = wsum.to(BF16).float()
                head = conv(name, stride, pad, w=torch.stack([hi, wsum - hi], 2))
            return _Conv(name, wp, pack_vec(sd[name + ".bias"], dev),
                         cin_pad or ci, co, (kt, kh, kw), stride, pad[0] if kh > 1 else 0, pad[1] if kh > 1 else 0, thin, frag,
                         head)

        def norm(name):
            return _Norm(pack_vec(sd[name + ".weight"], dev), pack_vec(sd[name + ".bias"], dev))

        def resnet(name):
            sc = conv(name + ".conv_shortcut", pad=(0, 0)) if (name + ".conv_shortcut.weight") in sd else None
            rb = _Resnet(norm(name + ".norm1"), conv(name + ".conv1"), norm(name + ".norm2"), conv(name + ".conv2"), sc)
            if sc is None and rb.conv1.cin != rb.conv2.cout:
                raise KeyError(f"{name}: width changes {rb.conv1.cin} -> {rb.conv2.cout} but {name}.conv_shortcut is missing "
                               "from the state dict (ResnetBlock3D, attn_video_vae.py:292-305)")
            return rb

        def attn(name):
            c = sd[name + ".to_q.weight"].shape[0]
            wqkv = torch.cat([sd[name + f".{k}.weight"] for k in ("to_q", "to_k", "to_v")], dim=0)
            bqkv = torch.cat([sd[name + f".{k}.bias"] for k in ("to_q", "to_k", "to_v")], dim=0)
            return _Attn(norm(name + ".group_norm"), pack_matrix(wqkv, dev), pack_vec(bqkv, dev),
                         pack_matrix(sd[name + ".to_out.0.weight"], dev), pack_vec(sd[name + ".to_out.0.bias"], dev), c)

        def mid(name):
            return (resnet(name + ".resnets.0"), attn(name + ".attentions.0"), resnet(name + ".resnets.1"))

        # ---- encoder (Encoder3D, attn_video_vae.py:671-856)
        self.attn_as_gemm = True            # mid-block attention as QK^T GEMM -> softmax -> PV GEMM (see _attention)
        self.enc_conv_in = conv("encoder.conv_in", cin_pad=4)
        self.enc_down = []
        for i in range(n):
            res = [resnet(f"encoder.down_blocks.{i}.resnets.{j}") for j in range(cfg.layers_per_block)]
            down = None
            if i != n - 1:
                temporal = i >= n - cfg.temporal_scale_num - 1
                down = conv(f"encoder.down_blocks.{i}.downsamplers.0.conv",
                            stride=(2 if temporal else 1, 2, 2), pad=(0, 1))   # F.pad(0,1,0,1) then pad-0 conv
            self.enc_down.append((res, down))
        self.enc_mid = mid("encoder.mid_block")
        self.enc_norm_out = norm("encoder.conv_norm_out")
        self.enc_conv_out = conv("encoder.conv_out")
        # ---- decoder (Decoder3D, attn_video_vae.py:859-1035)
        self.dec_conv_in = conv("decoder.conv_in")
        self.dec_mid = mid("decoder.mid_block")
        self.dec_up = []
        for i in range(n):
            res = [resnet(f"decoder.up_blocks.{i}.resnets.{j}") for j in range(cfg.layers_per_block + 1)]
            up = None
            if i != n - 1:
                u = f"decoder.up_blocks.{i}.upsamplers.0"
                w = sd[u + ".upscale_conv.weight"]
                up = _Up(pack_matrix(w.reshape(w.shape[0], w.shape[1]), dev), pack_vec(sd[u + ".upscale_conv.bias"], dev),
                         conv(u + ".conv"), i < cfg.temporal_scale_num, w.shape[1])
                if merge_upsamplers and w.shape[1] % 64 == 0:
                    # (weights as the reference holds them: cast to bf16 at load, model_loader.py:583-584; merged in fp32)
                    rz = 2 if up.temporal else 1
                    w1 = w.reshape(w.shape[0], w.shape[1]).to(device=dev, dtype=BF16)
                    b1 = sd[u + ".upscale_conv.bias"].to(device=dev, dtype=BF16)
                    w3, b3 = sd[u + ".conv.weight"].to(device=dev, dtype=BF16), sd[u + ".conv.bias"].to(device=dev, dtype=BF16)
                    up.merged = {}
                    # temporal: the three head patterns + the two steady-state parities; spatial-only: the steady-state window
                    # (the conv's own replicate-first padding serves the head)
                    for i in (range(5) if up.temporal else (w3.shape[2] - 1,)):
                        sig = subpixel.signature(i, rz, w3.shape[2])
                        if sig not in up.merged:
                            srcs, parts = subpixel.merge_upsampler(w1, b1, w3, b3, rz, sig)
                            packed = []
                            for py, px, wm, b, bb in parts:
                                wp = pack_conv3d(wm, dev)
                                # fragment-ordered copy for the sub-pixel conv kernel (weights streamed to registers)
                                frag = ops.pack_conv_frag(wp, len(srcs), w.shape[1], w3.shape[0], taps=(2, 2)) \
                                    if hasattr(ops, "pack_conv_frag") else None
                                packed.append((py, px, wp, b.contiguous(), bb.contiguous(), frag))
                            up.merged[sig] = (len(srcs), packed)
            self.dec_up.append((res, up))
        self.dec_norm_out = norm("decoder.conv_norm_out")
        self.dec_conv_out = conv("decoder.conv_out")
        self._iota = {}

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

    # ------------------------------------------------------------------ layer primitives
    def _conv(self, cw: _Conv, x: torch.Tensor, st: dict, first: bool, resid: Optional[torch.Tensor] = None,
              gn: bool = False, wide: bool = False):
        """Causal conv of one temporal slice.  ``gn=True``: also return the per-frame GroupNorm statistics of the
        output when the conv kernel can fuse them into its epilogue (else None) -> ``(out, stats)``.
        ``wide``: store the output in the trunk dtype (fp32 under ``trunk_fp32``) instead of the activation dtype."""
        ops = self.ops
        odt = wide if isinstance(wide, torch.dtype) else (self.trunk_dtype if wide else None)
        T, H, W, Cin = x.shape
        assert Cin == cw.cin, (cw.name, Cin, cw.cin)
        if first and cw.head is not None:
            return self._conv_causal_head(cw, x, st, resid, gn, odt)
        kt, kh, kw = cw.k
        sT, sH, sW = cw.stride
        carry = kt - sT                                     # frames handed to the next temporal slice
        halo = None if first else st.get(cw.name)
        pt = (kt - 1) if first else (halo.shape[0] if halo is not None else 0)
        if not first and carry > 0 and halo is None:
            raise RuntimeError(f"{cw.name}: missing temporal halo for a non-initial slice")
        assert (T + pt - kt) % sT == 0, (cw.name, T, pt)
        To = (T + pt - kt) // sT + 1
        Ho = (H + cw.pad_lo + cw.pad_hi - kh) // sH + 1
        Wo = (W + cw.pad_lo + cw.pad_hi - kw) // sW + 1
        geom = Conv3dGeom(T, H, W, Cin, To, Ho, Wo, cw.k, cw.stride, (pt, cw.pad_lo, cw.pad_lo), halo)
        if resid is not None and tuple(resid.shape) != (To, Ho, Wo, cw.cout):
            raise ValueError(f"{cw.name}: residual {tuple(resid.shape)} does not match the output {(To, Ho, Wo, cw.cout)}")
        out = ops.empty(To, Ho, Wo, cw.cout, dtype=odt)
        f32 = out.dtype in (torch.float32, torch.float16)      # (a wide store: fp32, or the h16 format -- ops derives the kind from out.dtype)
        K = cw.w.shape[1]
        epi = EPI_RESID_GATE if resid is not None else EPI_BIAS
        stats = None
        fused_thin = (cw.thin and cw.cin == 4 and tuple(cw.k[1:]) == (3, 3) and tuple(cw.stride) == (1, 1, 1) and K == 128
                      and cw.cout % 128 == 0 and (cw.pad_lo, cw.pad_hi) == (1, 1))
        if cw.thin and not fused_thin:                      # decoder conv_in (Cin 16, latent resolution): im2col + plain GEMM
            cols = ops.empty(To * Ho * Wo, K)
            ops.im2col_causal(x, cols, geom)
            ops.gemm(cols, cw.w, out, N=cw.cout, K=K, M=To * Ho * Wo, bias=cw.b, epilogue=epi, resid=resid,
                     lda=K, ldc=cw.cout, ldr=cw.cout, out_f32=f32)
        elif tuple(cw.k) == (1, 1, 1) and tuple(cw.stride) == (1, 1, 1) and cw.pad_lo == 0 and not gn and K == Cin:
            # a 1x1x1 conv (the ResnetBlock3D shortcuts, attn_video_vae.py:299-308) IS a plain GEMM over the NDHWC rows: no tap gather,
            # and the forms with N % 256 == 0 reach the persistent kernel (round 5, same box: -23 ms per config-3 step against the
            # implicit-GEMM route; these launches are HBM-bound either way)
            M_ = To * Ho * Wo
            ops.gemm(x.reshape(M_, Cin), cw.w, out.reshape(M_, cw.cout), N=cw.cout, K=K, M=M_, bias=cw.b, epilogue=epi,
                     resid=None if resid is None else resid.reshape(M_, cw.cout), out_f32=f32)
        else:
            # implicit-GEMM conv; RGB input (encoder conv_in, Cin 3 -> 4) is served by the thin-input variant of the
            # LDS-halo kernel, which builds the im2col image of each patch in LDS
            r = ops.gemm(x, cw.w, out, N=cw.cout, K=K, bias=cw.b, epilogue=epi, resid=resid, conv=geom,
                         ldc=cw.cout, ldr=cw.cout, gn_groups=self.cfg.norm_num_groups if gn else 0,
                         W_frag=None if cw.thin else cw.w_frag, out_f32=f32)
            stats = r[1] if gn else None
        if carry > 0 and not st.get("__last_slice__", False):   # per-conv memory for the next slice (none follows the last one)
            if T >= carry:
                st[cw.name] = x[T - carry:].clone()
            else:
                prev = halo if halo is not None else x[:1].expand(pt, H, W, Cin)
                st[cw.name] = torch.cat([prev, x], dim=0)[-carry:].contiguous()
        return (out, stats) if gn else out

    def _conv_causal_head(self, cw: _Conv, x, st, resid, gn, odt=None):
        """First slice of a clip through a kt = 3 stride-1 conv: output frame 0 = (hi + lo) * x[0] with hi + lo = W0+W1+W2
        (two taps on the replicated frame instead of three), frames 1.. = the plain conv over x with one replicated frame.
        Same outputs, per-frame statistics and carried state as one launch with two replicated frames."""
        T, H, W, _ = x.shape
        if resid is not None and tuple(resid.shape) != (T, H, W, cw.cout):
            raise ValueError(f"{cw.name}: residual {tuple(resid.shape)} does not match the output {(T, H, W, cw.cout)}")
        out = self.ops.empty(T, H, W, cw.cout, dtype=odt)
        stats = []
        for sub, n_in, o in ((cw.head, 1, 0), (cw, T, 1)):       # (weights, input frames 0..n_in-1, first output frame)
            To = n_in + 1 - sub.k[0] + 1
            if To <= 0:
                continue
            geom = Conv3dGeom(n_in, H, W, cw.cin, To, H, W, sub.k, (1, 1, 1), (1, cw.pad_lo, cw.pad_lo), None)
            r = resid[o:o + To] if resid is not None else None
            res = self.ops.gemm(x[:n_in], sub.w, out[o:o + To], N=cw.cout, K=sub.w.shape[1], bias=cw.b,
                                epilogue=EPI_RESID_GATE if r is not None else EPI_BIAS, resid=r, conv=geom,
                                ldc=cw.cout, ldr=cw.cout, gn_groups=self.cfg.norm_num_groups if gn else 0, W_frag=sub.w_frag,
                                out_f32=out.dtype in (torch.float32, torch.float16))
            if gn:
                stats.append(res[1])
        if not st.get("__last_slice__", False):
            carry = cw.k[0] - 1
            st[cw.name] = x[T - carry:].clone() if T >= carry else torch.cat([x[:1].expand(carry - T, H, W, cw.cin), x], 0).contiguous()
        if not gn:
            return out
        return out, (None if any(s is None for s in stats) else torch.cat(stats, 0))

    def _gn(self, nm: _Norm, x: torch.Tensor, silu: bool, stats: Optional[torch.Tensor] = None):
        """GroupNorm (+SiLU); ``stats`` = statistics of ``x`` already produced by the conv that wrote it."""
        ops, cfg = self.ops, self.cfg
        if stats is None:
            stats = ops.empty(x.shape[0], cfg.norm_num_groups, 2, dtype=torch.float64)
            ops.groupnorm_stats(x, stats, cfg.norm_num_groups)
        out = ops.empty(*x.shape)
        ops.groupnorm_apply(x, out, stats, nm.gamma, nm.beta, cfg.norm_num_groups, cfg.norm_eps, silu)
        return out

    def _resnet(self, rb: _Resnet, x, st, first, x_stats=None, wide=True):
        """-> (out, GroupNorm statistics of out or None).  ``x_stats``: statistics of ``x`` if its producer fused them.
        ``wide``: the output stays on the residual trunk (only GroupNorm and the next residual add read it) -> trunk dtype;
        False: a conv reads it as an MFMA operand -> activation dtype."""
        h = self._gn(rb.norm1, x, True, x_stats)
        h, hs = self._conv(rb.conv1, h, st, first, gn=True, wide=self.branch_dtype if self.branch_dtype is not None else False)
        h = self._gn(rb.norm2, h, True, hs)
        if rb.shortcut is not None and x.dtype != self.ops.act_dtype:
            raise RuntimeError(f"{rb.shortcut.name}: a shortcut conv reads its block input as an MFMA operand; the producer must store it "
                               f"in the activation dtype, got {x.dtype}")
        # (the shortcut's output is only ever the residual of conv2: a trunk tensor)
        sc = self._conv(rb.shortcut, x, st, first, wide=True) if rb.shortcut is not None else x
        return self._conv(rb.conv2, h, st, first, resid=sc, gn=True, wide=wide)

    def _attention(self, ab: _Attn, x):
        """Per-frame spatial self-attention of the mid block (1 head x C=512 over n = H*W tokens).
        n <= 65536: Q K^T as an MFMA GEMM with fp32 scores, row softmax, P V as a second GEMM, in blocks of 16384 query rows --
        1 KiB of K/V fragments per MFMA makes the fused single-pass kernel LDS-bound at head_dim 512, two plain
        GEMMs around a materialised score matrix (1 GiB fp32 per 1024-px tile frame, 4 GiB per block of an untiled 2048^2
        frame; 288 GB of HBM) are ~2x faster.  Larger n (untiled 4K frames) uses the fused variable-length kernel."""
        ops = self.ops
        T, H, W, Cc = x.shape
        n = H * W
        y = self._gn(ab.norm, x, False)
        out = ops.empty(T, H, W, Cc, dtype=self.trunk_dtype)
        if self.attn_as_gemm and n <= 65536 and n % 64 == 0:
            # query rows in blocks of at most 16384 (1 GiB of fp32 scores at a 1024-px tile; 4 GiB per block for the 65536-token
            # frames of an untiled 2048^2 clip -- BASELINE config 2 -- where the single-pass kernel is LDS-bound at head_dim 512)
            rb = min(n, 16384)
            npad = (n + 255) // 256 * 256
            q, v = ops.empty(T * n, Cc), ops.empty(T * n, Cc)
            k = ops.empty(T * n + npad, Cc)                               # slack: the GEMM reads whole 256-row W panels
            y2 = y.reshape(T * n, Cc)
            for dst, j in ((q, 0), (k, 1), (v, 2)):                       # (qkv_w rows are q | k | v blocks of C)
                ops.gemm(y2, ab.qkv_w[j * Cc:(j + 1) * Cc], dst[:T * n], N=Cc, K=Cc, bias=ab.qkv_b[j * Cc:(j + 1) * Cc].contiguous())
            S = ops.empty(rb, n, dtype=torch.float32)
            P = ops.empty(rb, n)
            att = ops.empty(T * n, Cc)
            for t in range(T):
                vt = v[t * n:(t + 1) * n].t().contiguous()               # layout only: V^T [C, n] is the K-contiguous W operand
                for r0 in range(0, n, rb):
                    r1 = min(r0 + rb, n)
                    ops.gemm(q[t * n + r0:t * n + r1], k[t * n:t * n + npad], S[:r1 - r0], N=n, K=Cc, out_f32=True)
                    ops.softmax_rows(S[:r1 - r0], P[:r1 - r0], 1.0 / math.sqrt(Cc))
                    ops.gemm(P[:r1 - r0], vt, att[t * n + r0:t * n + r1], N=Cc, K=n)
        else:
            qkv = ops.empty(T * n, 3 * Cc)
            ops.gemm(y.reshape(T * n, Cc), ab.qkv_w, qkv, N=3 * Cc, K=Cc, bias=ab.qkv_b)
            key = (T, n)
            if key not in self._iota:
                self._iota[key] = (torch.arange(T * n, dtype=torch.int32, device=self.device),
                                   (torch.arange(T + 1, dtype=torch.int32, device=self.device) * n).contiguous())
            rows, cu = self._iota[key]
            att = ops.empty(T * n, Cc)
            ops.attn_varlen(qkv, att, rows, rows, cu, n, 1, Cc, 1.0 / math.sqrt(Cc))
        ops.gemm(att, ab.out_w, out, N=Cc, K=Cc, M=T * n, bias=ab.out_b, epilogue=EPI_RESID_GATE,
                 resid=x, ldc=Cc, ldr=Cc, out_f32=out.dtype in (torch.float32, torch.float16))
        return out

    def _mid(self, m, x, st, first, x_stats=None, wide=True):
        x, _ = self._resnet(m[0], x, st, first, x_stats)
        x = self._attention(m[1], x)
        return self._resnet(m[2], x, st, first, wide=wide)

    def _upsample(self, up: _Up, x, st, first, wide=False):
        ops = self.ops
        T, H, W, Cc = x.shape
        if up.merged is not None:
            return self._upsample_subpixel(up, x, st, first, wide)
        rz = 2 if up.temporal else 1
        drop = up.temporal and first                       # remove_head on the first slice only
        To = T * rz - (1 if drop else 0)
        y = ops.empty(To, 2 * H, 2 * W, Cc)
        ops.gemm(x.reshape(T * H * W, Cc), up.upscale_w, y, N=4 * rz * Cc, K=Cc, M=T * H * W, bias=up.upscale_b,
                 ps=PixelShuffleGeom(T, H, W, rz, Cc, drop))
        return self._conv(up.conv, y, st, first, gn=True, wide=wide)

    def _upsample_subpixel(self, up: _Up, x, st, first, wide=False):
        """Upsampler as (kt', 2, 2)-tap convs over the low-resolution input, one launch per output phase, each scattering
        into its positions of the upsampled tensor (subpixel.py).  The causal memory of the reference's conv (the last frames
        of its upsampled input) becomes the last frame(s) of the LOW-resolution input, and the tap pattern of an output
        frame depends on its index in the whole clip, so the number of low-resolution frames consumed so far travels in
        the slice state.  -> (y, None): GroupNorm statistics are taken by the caller (these launches cannot fuse them)."""
        ops, cw = self.ops, up.conv
        T, H, W, Cc = x.shape
        rz = 2 if up.temporal else 1
        kt = cw.k[0]
        key = cw.name + "#subpixel"
        t0 = 0 if first else st.get(key + "#frames")
        mem = None if first else st.get(key)
        if not first and (t0 is None or mem is None):
            raise RuntimeError(f"{cw.name}: missing temporal state for a non-initial slice")
        carry = kt - 1 if rz == 1 else 1                   # low-resolution frames the next slice needs
        outs = [subpixel.output_frames(t0 + tl, rz) for tl in range(T)]
        y = ops.empty(sum(len(o) for o in outs), 2 * H, 2 * W, cw.cout, dtype=self.trunk_dtype if wide else None)
        f32 = y.dtype in (torch.float32, torch.float16)
        # GroupNorm statistics of y fused into the launches' epilogues where the ops offer it (one partial buffer for all of them)
        shared = {"frames": y.shape[0]} if hasattr(ops, "gn_shared_stats") else None

        def launch(sig, a, b, base, t_stride):
            """frames a..b-1 of this slice -> y[base + j * t_stride] with the merged weights of `sig`."""
            n_src, parts = up.merged[sig]
            pt = n_src - 1
            if pt == 0:
                halo = None
            elif a >= pt:
                halo = x[a - pt:a]
            elif mem is None:                                # head of the clip: replicate frame 0 (only rz = 1 gets here)
                halo = None if a == 0 else torch.cat([x[:1].expand(pt - a, H, W, Cc), x[:a]], 0).contiguous()
            else:
                halo = mem[mem.shape[0] - pt:] if a == 0 else torch.cat([mem[mem.shape[0] - (pt - a):], x[:a]], 0).contiguous()
            geom_in = x[a:b]
            if len(parts) == 4 and all(pp[5] is not None for pp in parts) and getattr(ops, "phase_quad", False):
                # all four spatial phases in ONE launch (svr_phase_scatter.quad): the phase is the fastest tile index, so the
                # four workgroups that stage the same low-resolution halo run side by side and three of them hit L2
                py, px, w, bias, bb, frag = parts[0]
                geom = Conv3dGeom(b - a, H, W, Cc, b - a, H, W, (n_src, 2, 2), (1, 1, 1), (pt, 1 - py, 1 - px), halo)
                kw = {}
                if shared is not None:
                    shared["frame0"] = base
                    kw = dict(gn_groups=self.cfg.norm_num_groups, gn_shared=shared)
                ops.gemm(geom_in, w, y[base:], N=cw.cout, K=w.shape[1], bias=bias, conv=geom,
                         phase=PhaseScatter(py, px, bb, t_stride, quad=list(parts)), W_frag=frag, out_f32=f32, **kw)
                return
            for py, px, w, bias, bb, frag in parts:
                geom = Conv3dGeom(b - a, H, W, Cc, b - a, H, W, (n_src, 2, 2), (1, 1, 1), (pt, 1 - py, 1 - px), halo)
                kw = {}
                if shared is not None:
                    shared["frame0"] = base
                    kw = dict(gn_groups=self.cfg.norm_num_groups, gn_shared=shared)
                ops.gemm(geom_in, w, y[base:], N=cw.cout, K=w.shape[1], bias=bias, conv=geom,
                         phase=PhaseScatter(py, px, bb, t_stride), W_frag=frag, out_f32=f32, **kw)

        if rz == 1:
            launch(subpixel.signature(kt - 1, 1, kt), 0, T, 0, 1)
        else:
            base, tl = 0, 0
            while tl < T:
                tg = t0 + tl
                if tg == 0:                                  # frame 0 of the clip: its single output frame
                    launch(subpixel.signature(0, 2, kt), tl, tl + 1, base, 1)
                    base, tl = base + 1, tl + 1
                elif tg == 1:                                # the two head patterns that still see frame 0's dropped sub-frame
                    launch(subpixel.signature(1, 2, kt), tl, tl + 1, base, 1)
                    launch(subpixel.signature(2, 2, kt), tl, tl + 1, base + 1, 1)
                    base, tl = base + 2, tl + 1
                else:                                        # steady state: both temporal phases of every remaining frame
                    launch(subpixel.signature(3, 2, kt), tl, T, base, 2)
                    launch(subpixel.signature(4, 2, kt), tl, T, base + 1, 2)
                    base, tl = base + 2 * (T - tl), T
        if not st.get("__last_slice__", False):
            st[key + "#frames"] = t0 + T
            if carry > 0:
                if T >= carry:
                    st[key] = x[T - carry:].clone()
                else:
                    prev = mem if mem is not None else x[:1].expand(carry, H, W, Cc)
                    st[key] = torch.cat([prev, x], dim=0)[-carry:].contiguous()
        return y, (ops.gn_shared_stats(shared) if shared is not None else None)

    # ------------------------------------------------------------------ one temporal slice through a network
    # (hs = GroupNorm statistics of h when the conv that produced h fused them into its epilogue, else None)
    # Trunk storage rule (trunk_fp32): a tensor on the skip path is wide (fp32) unless a conv reads it as an MFMA operand -- the
    # input of a down/upsampler, or of a block with a shortcut conv.
    def _encoder_slice(self, x, st, first):
        levels = self.enc_down
        feeds_shortcut = lambda i: i < len(levels) and levels[i][0][0].shortcut is not None
        h, hs = self._conv(self.enc_conv_in, x, st, first, gn=True, wide=not feeds_shortcut(0))
        for i, (res, down) in enumerate(levels):
            for j, rb in enumerate(res):
                h, hs = self._resnet(rb, h, st, first, hs, wide=not (down is not None and j == len(res) - 1))
            if down is not None:
                h, hs = self._conv(down, h, st, first, gn=True, wide=not feeds_shortcut(i + 1))
        h, hs = self._mid(self.enc_mid, h, st, first, hs)
        h = self._gn(self.enc_norm_out, h, True, hs)
        return self._conv(self.enc_conv_out, h, st, first)

    def _decoder_slice(self, z, st, first):
        h, hs = self._conv(self.dec_conv_in, z, st, first, gn=True, wide=True)
        levels = self.dec_up
        h, hs = self._mid(self.dec_mid, h, st, first, hs, wide=levels[0][0][0].shortcut is None)
        for i, (res, up) in enumerate(levels):
            for j, rb in enumerate(res):
                h, hs = self._resnet(rb, h, st, first, hs, wide=not (up is not None and j == len(res) - 1))
            if up is not None:
                h, hs = self._upsample(up, h, st, first, wide=levels[i + 1][0][0].shortcut is None)
                if i == self.cfg.temporal_scale_num - 1 and st.get("__keep__") is not None:
                    # full frame rate from here on, every layer causal in time: frames the caller will trim are not computed
                    # (only the clip's last slice can be cut: decode() already dropped the latent frames nobody needs)
                    done = st["__produced__"]
                    st["__produced__"] = done + h.shape[0]
                    n = max(1, min(h.shape[0], st["__keep__"] - done))
                    if n < h.shape[0]:
                        if not st.get("__last_slice__", False):   # (decode_clip trims the latents so that only the last slice is cut)
                            raise RuntimeError("keep_frames cuts a temporal slice that is not the clip's last one")
                        h, hs = h[:n], (hs[:n] if hs is not None else None)
        h = self._gn(self.dec_norm_out, h, True, hs)
        return self._conv(self.dec_conv_out, h, st, first, wide=self.sample_dtype if self.sample_dtype is not None else False)

    # ------------------------------------------------------------------ temporal slicing (slicing_encode / _decode)
    def _slices(self, T: int, unit: int, frames_per_slice: int) -> List[Tuple[int, int]]:
        """first slice = 1 + k*unit frames, then k*unit frames each (attn_video_vae.py:1254-1300 with k=1)."""
        if T <= 1 + frames_per_slice:
            return [(0, T)]
        k = max(unit, frames_per_slice // unit * unit)
        out = [(0, 1 + k)]
        s = 1 + k
        while s < T:
            out.append((s, min(s + k, T)))
            s += k
        return out

    def encode_clip(self, x_thwc: torch.Tensor, frames_per_slice: Optional[int] = None) -> torch.Tensor:
        """[T, H, W, 4] (RGB + zero pad) -> encoder output h [T', H/8, W/8, 2*latent] (mean || logvar)."""
        T, H, W, _ = x_thwc.shape
        assert T == 1 or T % 4 == 1, "clip length must be 4n+1 (VideoAutoencoderKLWrapper.preprocess)"
        if frames_per_slice is None:
            per_frame = H * W * self.cfg.block_out_channels[0] * 2
            frames_per_slice = max(4, int(self.act_budget_bytes // per_frame) // 4 * 4)
        st, outs = {}, []
        slices = self._slices(T, 4, frames_per_slice)
        for i, (a, b) in enumerate(slices):
            st["__last_slice__"] = i == len(slices) - 1      # its tail frames would never be read: skip the copies
            outs.append(self._encoder_slice(x_thwc[a:b], st, i == 0))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def decode_clip(self, z_thwc: torch.Tensor, latents_per_slice: Optional[int] = None,
                    keep_frames: Optional[int] = None) -> torch.Tensor:
        """[T', h, w, 16] -> [T, 8h, 8w, 3] (the first ``keep_frames`` frames of it: see decode)."""
        tf = self.cfg.temporal_downsample_factor
        if keep_frames is not None:
            if keep_frames < 1:
                raise ValueError("keep_frames must be >= 1")
            if keep_frames >= 1 + (z_thwc.shape[0] - 1) * tf:
                keep_frames = None                                   # nothing to trim
            else:                                                    # latent j > 0 first shows in output frame tf (j - 1) + 1:
                z_thwc = z_thwc[:(keep_frames - 1 + tf - 1) // tf + 1]     # drop the latents that only feed trimmed frames
        Tl, h, w, _ = z_thwc.shape
        if latents_per_slice is None:
            s = self.cfg.spatial_downsample_factor
            per_latent = (h * s) * (w * s) * 2 * self.cfg.block_out_channels[0] * 2 * self.cfg.temporal_downsample_factor
            latents_per_slice = max(1, int(self.act_budget_bytes // per_latent))
        st, outs = {"__keep__": keep_frames, "__produced__": 0}, []
        slices = self._slices(Tl, 1, latents_per_slice)
        for i, (a, b) in enumerate(slices):
            st["__last_slice__"] = i == len(slices) - 1
            outs.append(self._decoder_slice(z_thwc[a:b], st, i == 0))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    # ------------------------------------------------------------------ tile-level concurrency
    def _edge_dev(self, length: int, ov: int, fade_lo: bool, fade_hi: bool) -> torch.Tensor:
        """Blend weights of a tile edge on the device, cached per (length, overlap, faded sides): a host -> device copy of pageable
        memory inside the tile loop would block the host until the stream it is issued on has drained, which serialises the tiles
        (callers warm the cache before _run_tiles)."""
        key = (length, ov, fade_lo, fade_hi)
        w = self._edges.get(key)
        if w is None:
            w = self._edges[key] = _edge_weights(length, ov, fade_lo, fade_hi, _cos_ramp(ov)).to(self.device)
        return w

    def _run_tiles(self, jobs, compute, blend):
        """``compute(job)`` -> the tile's result (any number of launches on the CURRENT stream); ``blend(job, result)``
        accumulates it into the shared output.  Sequential on one stream when ``tile_streams`` <= 1 (and always on the CPU
        double of the C ABI).  Otherwise tile i is issued on side stream i mod n -- which first waits for everything the
        caller's stream has produced so far -- and its blend on the caller's stream behind the tile's completion event, in tile
        order: same launches with the same operands, same accumulation order, so the same bits.  The host never blocks; at most n
        tiles' activations are in flight (each side stream owns its allocator pool: 288 GB of HBM)."""
        n = min(self.tile_streams, len(jobs)) if self.device.type == "cuda" else 1
        if n <= 1:
            for job in jobs:
                blend(job, compute(job))
            return
        main = torch.cuda.current_stream(self.device)
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(device=self.device))
        ready = torch.cuda.Event()
        ready.record(main)
        for i, job in enumerate(jobs):
            side = self._streams[i % n]
            side.wait_event(ready)
            with torch.cuda.stream(side):
                result = compute(job)
                done = torch.cuda.Event()
                done.record(side)
            main.wait_event(done)
            blend(job, result)
            result.record_stream(main)              # (allocated on the side stream, last read by the blend on the caller's)
        for side in self._streams[:n]:              # the caller may free / overwrite the inputs once ITS stream is done
            done = torch.cuda.Event()
            done.record(side)
            main.wait_event(done)

    # ------------------------------------------------------------------ public API (runner level)
    def _to_thwc4(self, x_cthw: torch.Tensor) -> torch.Tensor:
        Cc, T, H, W = x_cthw.shape
        x = torch.zeros(T, H, W, 4, dtype=self.ops.act_dtype, device=self.device)
        x[..., :Cc] = x_cthw.to(device=self.device, dtype=self.ops.act_dtype).permute(1, 2, 3, 0)   # layout only
        return x

    @staticmethod
    def _resolve_act_budget(explicit: Optional[int], device, tile_streams: int) -> int:
        """explicit argument > env SVR_VAE_ACT_BUDGET_GIB > 24 GiB clamped to free device memory / 8 / tile_streams."""
        if explicit is not None:
            if int(explicit) <= 0:
                raise ValueError(f"act_budget_bytes must be positive, got {explicit}")
            return int(explicit)
        env = os.environ.get("SVR_VAE_ACT_BUDGET_GIB")
        if env:
            try:
                gib = float(env)
            except ValueError:
                raise ValueError(f"SVR_VAE_ACT_BUDGET_GIB must be a number of GiB, got {env!r}") from None
            if not (gib > 0 and math.isfinite(gib)):
                raise ValueError(f"SVR_VAE_ACT_BUDGET_GIB must be positive and finite, got {env!r}")
            return int(gib * (1 << 30))
        budget = 24 << 30
        dev = torch.device(device)
        if dev.type == "cuda" and torch.cuda.is_available():
            free, _total = torch.cuda.mem_get_info(dev)
            budget = min(budget, max(1 << 30, int(free) // 8 // max(1, int(tile_streams))))
        return budget

    def _guarded(self, call, inp: torch.Tensor):
        """Run ``call()``; if its result is not finite although its input ``inp`` was, and an h16 store is in use, once more with
        those stores in fp32 (see ``overflow_guard`` in __init__).  Cost: one fp32 reduction over the output and ONE HOST SYNC per
        encode() / decode() call (the pipeline makes one such call per temporal batch: microseconds against seconds); the input is
        only looked at when the output failed.  A non-finite INPUT (e.g. NaN latents from upstream) is not an h16 overflow: the call
        is not repeated, the non-finite result is returned as the reference would return it, with a warning naming the input."""
        out = call()
        if not self.overflow_guard or "h16" not in (self.trunk_store, self.branch_store):
            return out
        if bool(torch.isfinite(out.sum(dtype=torch.float32))):
            return out
        import warnings
        if not bool(torch.isfinite(inp.float().sum())):
            warnings.warn("VideoVAEEngine: the input of this call is not finite; its output is returned as computed (not an h16 "
                          "range problem, no fp32 re-run)", RuntimeWarning, stacklevel=3)
            return out
        warnings.warn("VideoVAEEngine: non-finite output with h16 stores (an activation beyond +-4.2e6?); repeating the call with "
                      "fp32 stores", RuntimeWarning, stacklevel=3)
        saved = (self.trunk_store, self.trunk_dtype, self.branch_store, self.branch_dtype)
        if self.trunk_store == "h16":
            self.trunk_store, self.trunk_dtype = "fp32", torch.float32
        if self.branch_store == "h16":
            self.branch_store, self.branch_dtype = "fp32", torch.float32
        self.overflow_reruns += 1
        try:
            return call()
        finally:
            self.trunk_store, self.trunk_dtype, self.branch_store, self.branch_dtype = saved

    @torch.no_grad()
    def encode(self, x_cthw: torch.Tensor, tiled: bool = False, tile_size=(512, 512), tile_overlap=(64, 64),
               frames_per_slice: Optional[int] = None) -> torch.Tensor:
        """[3, T, H, W] in [-1, 1] -> scaled latent [T', H/8, W/8, 16] = (mean - shift) * scale."""
        return self._guarded(lambda: self._encode(x_cthw, tiled, tile_size, tile_overlap, frames_per_slice), x_cthw)

    def _encode(self, x_cthw, tiled, tile_size, tile_overlap, frames_per_slice):
        cfg, ops = self.cfg, self.ops
        if x_cthw.dim() == 3:
            x_cthw = x_cthw.unsqueeze(1)
        x = self._to_thwc4(x_cthw)
        T, H, W, _ = x.shape
        s, lc = cfg.spatial_downsample_factor, cfg.latent_channels
        if not tiled or (H <= tile_size[0] and W <= tile_size[1]):
            hfull = self.encode_clip(x, frames_per_slice)
            out = ops.empty(hfull.shape[0], hfull.shape[1], hfull.shape[2], lc)
            return ops.affine_slice(hfull, out, cfg.scaling_factor, cfg.shifting_factor)
        lth, ltw = max(1, tile_size[0] // s), max(1, tile_size[1] // s)
        loh = max(0, min(tile_overlap[0] // s, lth - 1))
        low = max(0, min(tile_overlap[1] // s, ltw - 1))
        Hl, Wl = (H + s - 1) // s, (W + s - 1) // s
        buf = {}
        jobs = [(y0, y1, x0, x1) for (y0, y1) in _tile_ranges(Hl, lth, loh) for (x0, x1) in _tile_ranges(Wl, ltw, low)]

        def compute(job):
            y0, y1, x0, x1 = job
            tile = x[:, y0 * s:min(y1 * s, H), x0 * s:min(x1 * s, W)].contiguous()
            enc = self.encode_clip(tile, frames_per_slice)
            eh = min(y1 - y0, enc.shape[1], Hl - y0)
            ew = min(x1 - x0, enc.shape[2], Wl - x0)
            return enc if (eh, ew) == (enc.shape[1], enc.shape[2]) else enc[:, :eh, :ew].contiguous()

        def blend(job, enc):
            y0, y1, x0, x1 = job
            if not buf:
                buf["acc"] = torch.zeros(enc.shape[0], Hl, Wl, enc.shape[3], dtype=torch.float32, device=self.device)
                buf["cnt"] = torch.zeros(Hl, Wl, dtype=torch.float32, device=self.device)
            wy = self._edge_dev(enc.shape[1], loh, y0 > 0, y1 < Hl)
            wx = self._edge_dev(enc.shape[2], low, x0 > 0, x1 < Wl)
            ops.blend_accumulate(enc, buf["acc"], buf["cnt"], wy, wx, y0, x0)

        for (y0, y1, x0, x1) in jobs:                    # warm the weight cache (expected tile extents) before any tile is issued
            self._edge_dev(min(y1 - y0, Hl - y0), loh, y0 > 0, y1 < Hl)
            self._edge_dev(min(x1 - x0, Wl - x0), low, x0 > 0, x1 < Wl)
        self._run_tiles(jobs, compute, blend)
        acc, cnt = buf["acc"], buf["cnt"]
        out = ops.empty(acc.shape[0], Hl, Wl, lc)
        return ops.blend_finalize(acc, cnt, out, cfg.scaling_factor, cfg.shifting_factor)

    @torch.no_grad()
    def decode(self, latent_thwc: torch.Tensor, tiled: bool = False, tile_size=(512, 512), tile_overlap=(64, 64),
               latents_per_slice: Optional[int] = None, keep_frames: Optional[int] = None) -> torch.Tensor:
        """scaled latent [T', h, w, 16] -> sample [3, T, 8h, 8w] (or [3, 8h, 8w] for a single frame).
        ``keep_frames``: the caller keeps only the first n output frames (the pipeline trims the 4n+1 / uniform-batch padding
        after decode, generation_phases.py:953-958): the decoder is causal in time, so the latent frames that only feed
        trimmed output and, at full frame rate, the trimmed frames themselves are not computed -> [3, n, 8h, 8w], equal to
        the first n frames of the full decode."""
        return self._guarded(lambda: self._decode(latent_thwc, tiled, tile_size, tile_overlap, latents_per_slice, keep_frames),
                             latent_thwc)

    def _decode(self, latent_thwc, tiled, tile_size, tile_overlap, latents_per_slice, keep_frames):
        cfg, ops = self.cfg, self.ops
        lat = latent_thwc.to(device=self.device, dtype=self.ops.act_dtype).contiguous()
        if lat.dim() == 3:
            lat = lat.unsqueeze(0)
        if keep_frames is not None and keep_frames < 1:
            raise ValueError("keep_frames must be >= 1")
        Tl, H, W, lc = lat.shape
        z = ops.empty(Tl, H, W, lc)
        # latent / scale + shift  ==  (latent - (-shift*scale)) * (1/scale)
        ops.affine_slice(lat, z, 1.0 / cfg.scaling_factor, -cfg.shifting_factor * cfg.scaling_factor)
        s = cfg.spatial_downsample_factor
        lth, ltw = max(1, tile_size[0] // s), max(1, tile_size[1] // s)
        if not tiled or (H <= lth and W <= ltw):
            y = self.decode_clip(z, latents_per_slice, keep_frames)
        else:
            oh, ow = tile_overlap
            loh = max(0, min(oh // s, lth - 1))
            low = max(0, min(ow // s, ltw - 1))
            buf = {}
            jobs = [(y0, y1, x0, x1) for (y0, y1) in _tile_ranges(H, lth, loh) for (x0, x1) in _tile_ranges(W, ltw, low)]

            def compute(job):
                y0, y1, x0, x1 = job
                return self.decode_clip(z[:, y0:y1, x0:x1].contiguous(), latents_per_slice, keep_frames)

            def blend(job, dec):
                y0, y1, x0, x1 = job
                if not buf:
                    buf["acc"] = torch.zeros(dec.shape[0], H * s, W * s, dec.shape[3], dtype=torch.float32, device=self.device)
                    buf["cnt"] = torch.zeros(H * s, W * s, dtype=torch.float32, device=self.device)
                ho, wo = (y1 - y0) * s, (x1 - x0) * s
                ops.blend_accumulate(dec, buf["acc"], buf["cnt"], self._edge_dev(ho, oh, y0 > 0, y1 < H),
                                     self._edge_dev(wo, ow, x0 > 0, x1 < W), y0 * s, x0 * s)

            for (y0, y1, x0, x1) in jobs:                # warm the weight cache before any tile is issued
                self._edge_dev((y1 - y0) * s, oh, y0 > 0, y1 < H)
                self._edge_dev((x1 - x0) * s, ow, x0 > 0, x1 < W)
            self._run_tiles(jobs, compute, blend)
            acc, cnt = buf["acc"], buf["cnt"]
            y = ops.empty(*acc.shape, dtype=self.sample_dtype)
            ops.blend_finalize(acc, cnt, y, 1.0, 0.0)
        y = y.permute(3, 0, 1, 2)                           # layout only: [3, T, H, W]
        return y[:, 0] if y.shape[1] == 1 else y
