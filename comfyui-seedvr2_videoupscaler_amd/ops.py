"""Torch-tensor front end of the C ABI: validates layouts, passes raw device pointers and the current
HIP stream.  PyTorch is used for device memory and streams only; all arithmetic happens in
libseedvr2_hip.so.  Every method raises if the library is missing or a tensor is not on the GPU.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import hip_lib
from .hip_lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SILU, EPI_RESID_GATE, EPI_SWIGLU  # noqa: F401 (re-export)

BF16 = torch.bfloat16
# "h16": the 2-byte WIDE storage format of tensors that are not MFMA operands (SVR_STORE_H16 in include/seedvr2_hip.h): a
# torch.float16 tensor whose elements hold x * 2^-6.  Only the library's kernels interpret it (GroupNorm, residual adds);
# h16_to_float() is for tests and debugging.
H16 = torch.float16
H16_SCALE = 2.0 ** -6
STORE_BF16, STORE_FP32, STORE_H16 = 0, 1, 2
_KIND = {BF16: STORE_BF16, torch.float32: STORE_FP32, H16: STORE_H16}


def store_kind(t: torch.Tensor) -> int:
    try:
        return _KIND[t.dtype]
    except KeyError:
        raise ValueError(f"activation storage must be bf16, fp32 or h16 (torch.float16), got {t.dtype}") from None


def h16_to_float(t: torch.Tensor) -> torch.Tensor:
    return t.float() * (1.0 / H16_SCALE) if t.dtype == H16 else t.float()


@dataclass
class Conv3dGeom:
    """Causal conv geometry over an NDHWC input (see svr_conv_geom in include/seedvr2_hip.h)."""
    T: int
    H: int
    W: int
    Cin: int
    To: int
    Ho: int
    Wo: int
    k: tuple          # (kt, kh, kw)
    stride: tuple     # (st, sh, sw)
    pad: tuple        # (pt, ph, pw) front pads
    halo: Optional[torch.Tensor] = None   # [halo_frames, H, W, Cin] bf16


@dataclass
class PixelShuffleGeom:
    F: int
    H: int
    W: int
    rz: int
    C: int
    drop_first: bool


@dataclass
class PhaseScatter:
    """Sub-pixel convolution launch: the conv's output voxel (t, y, x) lands at out[t * t_stride, 2y + py, 2x + px] (``out``
    = the upsampled tensor from the launch's first frame on); ``bias_border`` fp32 [3, N]: bias on the voxels whose window
    loses its border tap (row border | column border | both)."""
    py: int
    px: int
    bias_border: Optional[torch.Tensor] = None
    t_stride: int = 1
    # quad launch (ABI v6): ALL four spatial phases in one launch -- a list of four (py, px, W, bias, bias_border, W_frag) in
    # any order; py / px / bias_border above and the call's own W / bias / W_frag / conv pads then describe phase (0, 0) only
    # as a fallback (HipOps.gemm issues the four launches itself when the library does not take the quad form)
    quad: Optional[list] = None


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def fill_gemm_args(A, W, out, *, N, K, M=None, bias=None, epilogue=EPI_BIAS, gate=None, resid=None, out_f32=False,
                   conv: Optional[Conv3dGeom] = None, ps: Optional[PixelShuffleGeom] = None, lda=None, ldc=None, ldr=None,
                   W_frag=None, phase: Optional[PhaseScatter] = None, zeros_ptr: int = 0, chk=None, ptr=None):
    """svr_gemm_args for one ``gemm`` call (layout checks included).  ``chk(tensor, dtype, name)`` validates device / dtype /
    contiguity (HipOps._chk), ``ptr(tensor)`` yields the device address -- both injectable so that tools and CPU tests can fill
    the struct for shape-only (meta) tensors and ask the library which kernel would serve the call (svr_gemm_kernel_class)."""
    chk = chk or (lambda t, dtype=None, name="tensor": t)
    ptr = ptr or (lambda t: t.data_ptr())
    chk(A, BF16, "A"); chk(W, BF16, "W")
    # ``out_f32``: the caller expects a wide store (fp32, or h16 when ``out`` is a torch.float16 tensor); the kind itself comes
    # from out.dtype
    okind = store_kind(out)
    if bool(out_f32) != (okind != STORE_BF16):
        raise ValueError(f"out must be {'fp32 / h16' if out_f32 else 'bf16'}, got {out.dtype}")
    chk(out, None, "out")
    if W.shape[1] != K or W.shape[0] < N or W.shape[0] % 128:
        raise ValueError(f"packed weight shape {tuple(W.shape)} incompatible with N={N}, K={K}")
    a = hip_lib.GemmArgs()
    if conv is not None:
        M = conv.To * conv.Ho * conv.Wo
        g = a.conv
        g.enabled = 1
        g.T, g.H, g.W, g.Cin = conv.T, conv.H, conv.W, conv.Cin
        g.To, g.Ho, g.Wo = conv.To, conv.Ho, conv.Wo
        g.kt, g.kh, g.kw = conv.k
        g.st, g.sh, g.sw = conv.stride
        g.pt, g.ph, g.pw = conv.pad
        if conv.halo is not None:
            chk(conv.halo, BF16, "halo")
            g.halo_frames = conv.halo.shape[0]
            g.halo = ptr(conv.halo)
        g.zeros = zeros_ptr
        if A.numel() != conv.T * conv.H * conv.W * conv.Cin:
            raise ValueError("conv input size mismatch")
        lda = 0
    else:
        if M is None:
            M = A.shape[0]
        if lda is None:
            lda = A.stride(0) if A.dim() == 2 else K
    if phase is not None:
        if conv is None or ps is not None or resid is not None:
            raise ValueError("phase scatter: conv mode only, without pixel shuffle / residual; fused statistics through gn_shared")
        if out.numel() < ((conv.To - 1) * phase.t_stride + 1) * 4 * conv.Ho * conv.Wo * N or not out.is_contiguous():
            raise ValueError("phase scatter: out must be the dense [frames, 2*Ho, 2*Wo, N] tensor from the launch's first frame on")
        a.phase.enabled, a.phase.py, a.phase.px, a.phase.t_stride = 1, int(phase.py), int(phase.px), int(phase.t_stride)
        if phase.quad is not None:
            if len(phase.quad) != 4 or sorted((q[0], q[1]) for q in phase.quad) != [(0, 0), (0, 1), (1, 0), (1, 1)]:
                raise ValueError("phase scatter: quad needs the four phases (0,0) (0,1) (1,0) (1,1)")
            a.phase.quad = 1
            for qpy, qpx, qw, qb, qbb, qfrag in phase.quad:
                i = 2 * int(qpy) + int(qpx)
                if qfrag is None or qfrag.numel() < N * K or (qbb is not None and tuple(qbb.shape) != (3, N)):
                    raise ValueError("phase scatter: every quad phase needs its fragment-ordered weights (and a [3, N] border bias)")
                a.phase.W_frag4[i] = ptr(chk(qfrag, BF16, "W_frag"))
                a.phase.bias4[i] = None if qb is None else ptr(chk(qb, torch.float32, "bias"))
                a.phase.bias_border4[i] = None if qbb is None else ptr(chk(qbb, torch.float32, "bias_border"))
        if phase.bias_border is not None:
            if tuple(phase.bias_border.shape) != (3, N):
                raise ValueError("phase scatter: bias_border must be [3, N]")
            a.phase.bias_border = ptr(chk(phase.bias_border, torch.float32, "bias_border"))
        ldc = 0
    if ps is not None:
        a.ps.enabled = 1
        a.ps.F, a.ps.H, a.ps.W, a.ps.rz, a.ps.C = ps.F, ps.H, ps.W, ps.rz, ps.C
        a.ps.drop_first = int(ps.drop_first)
        ldc = 0
    elif ldc is None:
        ldc = out.stride(0) if out.dim() == 2 else (N // 2 if epilogue == EPI_SWIGLU else N)
    a.A, a.lda, a.W, a.C, a.ldc = ptr(A), lda, ptr(W), ptr(out), ldc
    a.M, a.N, a.K = M, N, K
    if bias is not None:
        a.bias = ptr(chk(bias, torch.float32, "bias"))
    if gate is not None:
        a.gate = ptr(chk(gate, torch.float32, "gate"))
    if resid is not None:
        chk(resid, None, "resid")
        a.resid = ptr(resid)
        a.resid_f32 = store_kind(resid)
        a.ldr = ldr if ldr is not None else (resid.stride(0) if resid.dim() == 2 else N)
    a.epilogue, a.out_f32 = epilogue, okind
    if W_frag is not None:
        if W_frag.numel() < N * K:
            raise ValueError(f"W_frag holds {W_frag.numel()} elements, the problem needs N * K = {N * K}")
        a.W_frag = ptr(chk(W_frag, BF16, "W_frag"))
    return a, M


class HipOps:
    """The product backend.  One instance per device."""

    name = "hip"
    act_dtype = BF16          # storage dtype of activations
    phase_quad = True         # gemm(phase=PhaseScatter(quad=[...])): the four phases of a sub-pixel upsampler in one launch

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise hip_lib.HipLibraryError("HipOps needs a ROCm GPU device (cuda:N); there is no CPU fallback")
        self.lib = hip_lib.lib()          # raises loudly if the .so is not built
        with torch.cuda.device(self.device):
            buf = C.create_string_buffer(256)
            rc = self.lib.svr_device_info(buf, 256)
            self.device_info = buf.value.decode()
            if rc != 0:
                raise hip_lib.HipLibraryError(f"unsupported device: {self.device_info}")
        self.zeros = torch.zeros(64, dtype=torch.uint8, device=self.device)
        self.record_kernel_class, self.last_kernel_class = False, None

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t, dtype=None, name="tensor"):
        if t.device != self.device and not (t.device.type == "cuda" and t.device.index == (self.device.index or 0)):
            raise ValueError(f"{name} must live on {self.device}, got {t.device}")
        if dtype is not None and t.dtype != dtype:
            raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
        return t

    def _opt(self, t, dtype, name, numel=None):
        """Optional (or small side) operand: None -> NULL, else device / dtype / contiguity (and element count) checked."""
        if t is None:
            return None
        self._chk(t, dtype, name)
        if numel is not None and t.numel() != numel:
            raise ValueError(f"{name} must hold {numel} elements, got {tuple(t.shape)}")
        return C.c_void_p(t.data_ptr())

    def mfma_calibrate(self, seconds: float = 0.25, reps: int = 3) -> float:
        """Measurement aid (svr_mfma_calibrate; never on the data path): TFLOP/s of a bare MFMA loop on this device right now =
        the matrix-pipe rate its power limit allows.  A short launch sizes the iteration count for ``seconds``, then the median of
        ``reps`` launches timed with events on the current stream is returned."""
        with torch.cuda.device(self.device):
            ws = torch.empty(int(self.lib.svr_mfma_calibrate_workspace_bytes()), dtype=torch.uint8, device=self.device)
            flops = C.c_double(0.0)

            def run(iters):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                hip_lib.check(self.lib.svr_mfma_calibrate(C.c_void_p(ws.data_ptr()), int(iters), C.byref(flops), self._stream()),
                              "svr_mfma_calibrate")
                e.record()
                e.synchronize()
                return flops.value / (s.elapsed_time(e) * 1e-3) / 1e12

            run(2000)                                            # (first launch: code-object load)
            rate = run(20000)
            iters = max(2000, int(20000 * seconds / (flops.value / (rate * 1e12))))
            return sorted(run(iters) for _ in range(reps))[reps // 2]

    def set_option(self, key: str, value: int):
        """Tuning / measurement knob of the library (see svr_set_option in include/seedvr2_hip.h)."""
        hip_lib.check(self.lib.svr_set_option(key.encode(), int(value)), "svr_set_option")
        hip_lib.OPTIONS[key] = int(value)

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or BF16, device=self.device)

    # ------------------------------------------------------------------ GEMM / conv
    def pack_conv_frag(self, W, kt: int, Cin: int, N: int, taps=(3, 3)):
        """Fragment-ordered copy of packed conv weights W [Npad, kt*kh*kw*Cin] for gemm(..., W_frag=) (3x3 spatial taps for
        the LDS-halo kernel, 2x2 for the sub-pixel upsampler kernel; stride 1, Cin % 32 == 0, N % 128 == 0); None when the
        geometry is not served by those kernels."""
        kh, kw = taps
        if Cin % 64 or N % 128 or W.shape[1] != kt * kh * kw * Cin or (kh, kw) not in ((3, 3), (2, 2)):
            return None
        self._chk(W, BF16, "W")
        out = torch.empty(N * W.shape[1], dtype=BF16, device=self.device)
        hip_lib.check(self.lib.svr_conv_pack_frag_taps(_ptr(W), _ptr(out), N, W.shape[1], kt, kh, kw, Cin, self._stream()),
                      "svr_conv_pack_frag_taps")
        return out

    def pack_gemm_frag(self, W):
        """Fragment-ordered copy of a packed plain-GEMM weight W [Npad, K] for gemm(..., W_frag=): the persistent GEMM kernel then
        streams the weights straight into registers and only the activations pass through LDS (csrc/svr_gemm_w4r.hip; same bits).
        None for shapes that kernel never serves (it wants whole 256-column tiles and at least two 64-wide K tiles)."""
        n, k = W.shape
        if n % 256 or k % 64 or k < 128:
            return None
        self._chk(W, BF16, "W")
        out = torch.empty(n * k, dtype=BF16, device=self.device)
        hip_lib.check(self.lib.svr_gemm_pack_frag(_ptr(W), _ptr(out), n, k, self._stream()), "svr_gemm_pack_frag")
        return out

    def gemm(self, A, W, out, *, N, K, M=None, bias=None, epilogue=EPI_BIAS, gate=None, resid=None,
             out_f32=False, conv: Optional[Conv3dGeom] = None, ps: Optional[PixelShuffleGeom] = None,
             lda=None, ldc=None, ldr=None, gn_groups: int = 0, W_frag=None, phase: Optional[PhaseScatter] = None,
             gn_shared: Optional[dict] = None):
        """out[M, N] = A[M, K] @ W[:N, :K]^T with fused epilogue.  W is [Npad, K] bf16 (Npad % 128 == 0).
        With ``gn_groups`` > 0 (conv mode) returns ``(out, stats)``: per-frame GroupNorm (sum, sumsq) of the stored
        output [To, groups, 2] fp64 fused into the conv epilogue, or ``None`` when the kernel serving this
        geometry does not produce them (the caller then runs groupnorm_stats).
        ``gn_shared`` (with ``gn_groups``): several launches write ONE output tensor (the phases of a sub-pixel upsampler); the
        caller passes the same dict {"frames": output frames, "frame0": first output frame of this launch} to each of them and
        calls gn_shared_stats() after the last -- the launches then return ``out`` only."""
        if phase is not None and gn_groups and gn_shared is None:
            raise ValueError("phase scatter: fused statistics through gn_shared")
        if phase is not None and phase.quad is not None and not self._quad_ok(A, W, out, N, K, conv, phase, out_f32):
            # the library does not take this geometry as one launch: four phase launches (same results, same order)
            import dataclasses
            for qpy, qpx, qw, qb, qbb, qfrag in sorted(phase.quad, key=lambda q: (q[0], q[1])):
                g1 = dataclasses.replace(conv, pad=(conv.pad[0], 1 - qpy, 1 - qpx))
                self.gemm(A, qw, out, N=N, K=K, bias=qb, conv=g1, phase=PhaseScatter(qpy, qpx, qbb, phase.t_stride), W_frag=qfrag,
                          out_f32=out_f32, gn_groups=gn_groups, gn_shared=gn_shared)
            return out
        a, M = fill_gemm_args(A, W, out, N=N, K=K, M=M, bias=bias, epilogue=epilogue, gate=gate, resid=resid, out_f32=out_f32,
                              conv=conv, ps=ps, lda=lda, ldc=ldc, ldr=ldr, W_frag=W_frag, phase=phase,
                              zeros_ptr=self.zeros.data_ptr(), chk=self._chk)
        stats = None
        if gn_groups > 0 and conv is not None:
            a.gn_groups = gn_groups
            nblk = int(self.lib.svr_gemm_gn_blocks(C.byref(a)))
            if gn_shared is not None:
                # one partial buffer [frames][nblk][groups] for all launches of the output tensor; a launch that cannot fuse
                # (nblk == 0, or a different block count than its siblings) switches the whole tensor back to the separate pass
                if nblk > 0 and gn_shared.get("ok", True) and gn_shared.get("nblk", nblk) == nblk:
                    if "partial" not in gn_shared:
                        gn_shared.update(nblk=nblk, groups=gn_groups, ok=True, partial=torch.empty(
                            gn_shared["frames"] * nblk * gn_groups * 2, dtype=torch.float64, device=self.device))
                    a.gn_partial = gn_shared["partial"].data_ptr() + int(gn_shared["frame0"]) * nblk * gn_groups * 16
                else:
                    gn_shared["ok"] = False
                    a.gn_groups = 0
            elif nblk > 0:
                partial = torch.empty(conv.To * nblk * gn_groups * 2, dtype=torch.float64, device=self.device)
                a.gn_partial = partial.data_ptr()
                stats = torch.empty(conv.To, gn_groups, 2, dtype=torch.float64, device=self.device)
            else:
                a.gn_groups = 0
        if self.record_kernel_class:                      # (bench.py: which kernel does the library pick for this launch)
            self.last_kernel_class = hip_lib.KERNEL_CLASSES.get(int(self.lib.svr_gemm_kernel_class(C.byref(a))), "invalid")
        hip_lib.check(self.lib.svr_gemm_bf16(C.byref(a), self._stream()), "svr_gemm_bf16")
        if gn_groups > 0 and gn_shared is None:
            if stats is not None:
                hip_lib.check(self.lib.svr_groupnorm_reduce(_ptr(partial), _ptr(stats), conv.To, nblk, gn_groups,
                                                            self._stream()), "svr_groupnorm_reduce")
            return out, stats
        return out

    def _quad_ok(self, A, W, out, N, K, conv, phase, out_f32) -> bool:
        """Does the library serve this quad phase launch with the sub-pixel conv kernel (svr_gemm_kernel_class)?"""
        if any(q[5] is None for q in phase.quad):
            return False
        a, _ = fill_gemm_args(A, W, out, N=N, K=K, conv=conv, phase=phase, out_f32=out_f32, zeros_ptr=self.zeros.data_ptr(),
                              chk=self._chk)
        return hip_lib.KERNEL_CLASSES.get(int(self.lib.svr_gemm_kernel_class(C.byref(a)))) == "conv_subpixel"

    def gn_shared_stats(self, gn_shared: dict):
        """Statistics [frames, groups, 2] of a tensor whose launches shared ``gn_shared`` (gemm), or None when one of them could
        not fuse them (the caller then runs groupnorm_stats)."""
        if not gn_shared.get("ok", False) or "partial" not in gn_shared:
            return None
        stats = torch.empty(gn_shared["frames"], gn_shared["groups"], 2, dtype=torch.float64, device=self.device)
        hip_lib.check(self.lib.svr_groupnorm_reduce(_ptr(gn_shared["partial"]), _ptr(stats), gn_shared["frames"], gn_shared["nblk"],
                                                    gn_shared["groups"], self._stream()), "svr_groupnorm_reduce")
        return stats

    # ------------------------------------------------------------------ DiT side kernels
    def _xf32(self, x, name="x"):
        """activation inputs that may come from the wide residual trunk: -> the SVR_STORE_* kind (0 bf16, 1 fp32, 2 h16)"""
        self._chk(x, None, name)
        return store_kind(x)

    def rmsnorm_mod(self, x, out, eps, w=None, scale=None, shift=None):
        xf = self._xf32(x); self._chk(out, BF16, "out")
        rows, dim = x.shape
        if tuple(out.shape) != (rows, dim):
            raise ValueError("rmsnorm_mod: x must be bf16, fp32 or h16 [rows, dim] and out bf16 of the same shape")
        hip_lib.check(self.lib.svr_rmsnorm_mod(_ptr(x), _ptr(out), rows, dim, eps, self._opt(w, torch.float32, "w", dim),
                                               self._opt(scale, torch.float32, "scale", dim),
                                               self._opt(shift, torch.float32, "shift", dim), xf,
                                               self._stream()), "svr_rmsnorm_mod")
        return out

    def ada_combine(self, emb, params, slots, out):
        self._chk(emb, BF16, "emb"); self._chk(params, BF16, "params")
        self._chk(slots, torch.int32, "slots"); self._chk(out, torch.float32, "out")
        n_vec, dim = params.shape
        hip_lib.check(self.lib.svr_ada_combine(_ptr(emb), _ptr(params), _ptr(slots), _ptr(out), n_vec, dim,
                                               self._stream()), "svr_ada_combine")
        return out

    def qknorm_rope(self, qkv, heads, pos, t_offset, cos_tab, sin_tab, wq, wk, eps):
        self._chk(qkv, BF16, "qkv"); self._chk(pos, torch.int16, "pos")
        self._chk(cos_tab, torch.float32, "cos_tab"); self._chk(sin_tab, torch.float32, "sin_tab")
        rows = qkv.shape[0]
        if qkv.shape[1] != 3 * heads * 128 or pos.shape != (rows, 3):
            raise ValueError("qknorm_rope: bad shapes")
        n_pos, n_freq = cos_tab.shape
        if sin_tab.shape != cos_tab.shape:
            raise ValueError("qknorm_rope: cos / sin tables of different shapes")
        if wq is None or wk is None:
            raise ValueError("qknorm_rope: the q / k norm weights are required (fp32 [128] each)")
        hip_lib.check(self.lib.svr_qknorm_rope(_ptr(qkv), rows, heads, _ptr(pos), t_offset, _ptr(cos_tab),
                                               _ptr(sin_tab), n_pos, n_freq, self._opt(wq, torch.float32, "wq", 128),
                                               self._opt(wk, torch.float32, "wk", 128), eps,
                                               self._stream()), "svr_qknorm_rope")
        return qkv

    def attn_varlen(self, qkv, out, seq_rows, out_rows, cu, max_len, heads, head_dim, scale):
        self._chk(qkv, BF16, "qkv"); self._chk(out, BF16, "out")
        for t, n in ((seq_rows, "seq_rows"), (out_rows, "out_rows"), (cu, "cu")):
            self._chk(t, torch.int32, n)
        if qkv.dim() != 2 or out.dim() != 2 or qkv.shape[1] != 3 * heads * head_dim or out.shape[1] != heads * head_dim or \
                seq_rows.numel() != out_rows.numel() or cu.numel() < 1:
            raise ValueError("attn_varlen: qkv [rows, 3 * heads * head_dim], out [rows', heads * head_dim], one out_row per seq_row")
        hip_lib.check(self.lib.svr_attn_varlen(_ptr(qkv), qkv.stride(0), _ptr(out), out.stride(0), _ptr(seq_rows),
                                               _ptr(out_rows), _ptr(cu), cu.numel() - 1, max_len, heads, head_dim,
                                               scale, self._stream()), "svr_attn_varlen")
        return out

    def softmax_rows(self, S, P, scale):
        """P[r] = softmax(scale * S[r]); S fp32 [rows, cols], P bf16 [rows, cols]."""
        self._chk(S, torch.float32, "S"); self._chk(P, BF16, "P")
        rows, cols = S.shape
        hip_lib.check(self.lib.svr_softmax_rows(_ptr(S), _ptr(P), rows, cols, S.stride(0), P.stride(0), scale,
                                                self._stream()), "svr_softmax_rows")
        return P

    def rows_mean(self, src, dst, n_groups, rows_per_group):
        self._chk(src, BF16, "src"); self._chk(dst, BF16, "dst")
        if src.numel() != n_groups * rows_per_group * src.shape[-1] or dst.numel() != rows_per_group * src.shape[-1]:
            raise ValueError("rows_mean: src [n_groups * rows_per_group, dim], dst [rows_per_group, dim]")
        hip_lib.check(self.lib.svr_rows_mean(_ptr(src), _ptr(dst), n_groups, rows_per_group, src.shape[-1],
                                             self._stream()), "svr_rows_mean")
        return dst

    def patchify(self, vid, out):
        self._chk(vid, BF16, "vid"); self._chk(out, BF16, "out")
        T, H, W, Cc = vid.shape
        hip_lib.check(self.lib.svr_patchify(_ptr(vid), _ptr(out), T, H, W, Cc, out.shape[1], self._stream()),
                      "svr_patchify")
        return out

    def unpatchify_euler(self, pred, x_t, out):
        self._chk(pred, BF16, "pred"); self._chk(out, BF16, "out")
        T, H, W, Cc = out.shape
        if x_t is not None:
            self._chk(x_t, BF16, "x_t")
        hip_lib.check(self.lib.svr_unpatchify_euler(_ptr(pred), pred.stride(0), _ptr(x_t), _ptr(out), T, H, W, Cc,
                                                    self._stream()), "svr_unpatchify_euler")
        return out

    # ------------------------------------------------------------------ VAE side kernels
    def groupnorm_stats(self, x, stats, groups):
        xf = self._xf32(x); self._chk(stats, torch.float64, "stats")
        T, H, W, Cc = x.shape
        if stats.numel() != T * groups * 2:
            raise ValueError("groupnorm_stats: stats must be [T, groups, 2]")
        ws = torch.empty(int(self.lib.svr_groupnorm_workspace_bytes(T, H * W, groups)), dtype=torch.uint8,
                         device=self.device)
        hip_lib.check(self.lib.svr_groupnorm_stats(_ptr(x), _ptr(stats), _ptr(ws), T, H * W, Cc, groups, xf,
                                                   self._stream()), "svr_groupnorm_stats")
        return stats

    def groupnorm_apply(self, x, out, stats, gamma, beta, groups, eps, silu):
        xf = self._xf32(x); self._chk(out, BF16, "out")
        T, H, W, Cc = x.shape
        if out.numel() != x.numel():
            raise ValueError("groupnorm_apply: out must have x's shape")
        if stats is None or gamma is None or beta is None:
            raise ValueError("groupnorm_apply: stats, gamma and beta are required")
        hip_lib.check(self.lib.svr_groupnorm_apply(_ptr(x), _ptr(out), self._opt(stats, torch.float64, "stats", T * groups * 2),
                                                   self._opt(gamma, torch.float32, "gamma", Cc),
                                                   self._opt(beta, torch.float32, "beta", Cc), T, H * W,
                                                   Cc, groups, eps, int(silu), xf, self._stream()), "svr_groupnorm_apply")
        return out

    def im2col_causal(self, x, out, conv: Conv3dGeom):
        self._chk(x, BF16, "x"); self._chk(out, BF16, "out")
        g = hip_lib.ConvGeom()
        g.enabled = 1
        g.T, g.H, g.W, g.Cin = conv.T, conv.H, conv.W, conv.Cin
        g.To, g.Ho, g.Wo = conv.To, conv.Ho, conv.Wo
        g.kt, g.kh, g.kw = conv.k
        g.st, g.sh, g.sw = conv.stride
        g.pt, g.ph, g.pw = conv.pad
        if conv.halo is not None:
            g.halo_frames = conv.halo.shape[0]
            g.halo = self._chk(conv.halo, BF16, "halo").data_ptr()
        g.zeros = self.zeros.data_ptr()
        hip_lib.check(self.lib.svr_im2col_causal(_ptr(x), _ptr(out), C.byref(g), out.shape[1], self._stream()),
                      "svr_im2col_causal")
        return out

    def blend_accumulate(self, tile, acc, cnt, wy, wx, y0, x0):
        self._chk(tile, BF16, "tile"); self._chk(acc, torch.float32, "acc"); self._chk(cnt, torch.float32, "cnt")
        T, h, w, Cc = tile.shape
        Ta, H, W, Ca = acc.shape
        if (Ta, Ca) != (T, Cc) or cnt.numel() != H * W:
            raise ValueError("blend_accumulate: acc [T, H, W, C] for tile [T, h, w, C], cnt [H, W]")
        hip_lib.check(self.lib.svr_blend_accumulate(_ptr(tile), _ptr(acc), _ptr(cnt), self._opt(wy, torch.float32, "wy", h),
                                                    self._opt(wx, torch.float32, "wx", w), T, h, w, Cc,
                                                    H, W, y0, x0, self._stream()), "svr_blend_accumulate")

    def blend_finalize(self, acc, cnt, out, scale=1.0, shift=0.0):
        self._chk(acc, torch.float32, "acc"); self._chk(out, BF16, "out")
        T, H, W, Cc = acc.shape
        self._chk(cnt, torch.float32, "cnt")
        if cnt.numel() != H * W or out.numel() != T * H * W * out.shape[-1]:
            raise ValueError("blend_finalize: cnt [H, W], out [T, H, W, c_take]")
        hip_lib.check(self.lib.svr_blend_finalize(_ptr(acc), _ptr(cnt), _ptr(out), T, H * W, Cc, out.shape[-1], scale,
                                                  shift, self._stream()), "svr_blend_finalize")
        return out

    def affine_slice(self, inp, out, scale=1.0, shift=0.0):
        self._chk(inp, BF16, "inp"); self._chk(out, BF16, "out")
        c_in, c_out = inp.shape[-1], out.shape[-1]
        rows = inp.numel() // c_in
        hip_lib.check(self.lib.svr_affine_slice(_ptr(inp), _ptr(out), rows, c_in, c_out, scale, shift, self._stream()),
                      "svr_affine_slice")
        return out
