"""-m gpu: per-kernel numerics of the HIP C ABI against the plain-PyTorch fp32 reference of the same op
(tests/ops_reference.py), same bf16-rounded inputs.

Tolerances (stated here as the contract):
  * fp32-store test epilogue (``out_f32``): rel-err <= 1e-3 vs the fp32 reference (north-star bound);
  * production bf16 store: rel-err <= 2.5e-3 (one bf16 rounding of a perfect result is already 1.7e-3);
  * integer/index kernels (patchify, im2col): bit exact.
rel-err = ||a - b||_2 / ||b||_2.
"""
import math

import pytest
import torch

from conftest import sub, rel_err
from ops_reference import TorchOps, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SILU, EPI_RESID_GATE, EPI_SWIGLU

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
TOL_BF16 = 2.5e-3
TOL_F32 = 1e-3


@pytest.fixture(scope="module")
def hip():
    return sub("ops").HipOps("cuda:0")


@pytest.fixture(scope="module")
def ref():
    return TorchOps("cuda:0", act_dtype=torch.float32)


@pytest.fixture(params=[1, 2], ids=["epi_direct", "epi_lds"])
def gemm_epi(request, hip):
    """Both epilogues of svr::gemm_kernel on the same problem: 1 = stores straight from the accumulator layout, 2 = through
    LDS with row-contiguous 16-byte stores wherever the layout allows it (the library picks per problem by default)."""
    hip.set_option("gemm_epi", request.param)
    yield request.param
    hip.set_option("gemm_epi", 0)


def rnd(*shape, scale=1.0, seed=0, dtype=BF16):
    g = torch.Generator(device="cuda").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(dtype)


def packed(n, k, seed=1):
    packing = sub("packing")
    w = rnd(n, k, scale=1.0 / math.sqrt(k), seed=seed)
    return w, packing.pack_matrix(w, "cuda")


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 768, 256), (257, 64, 192), (1, 2560, 256),
                                   (513, 384, 64), (2048, 1536, 2560),
                                   # 256-wide N tiles: 1, 3 and 5 K tiles, ragged M
                                   (300, 256, 64), (700, 512, 192), (255, 512, 320), (58, 2560, 5120),
                                   # >= 256 tiles of 256x256: the wide-tile kernel (fewer tiles run 256x128 so no CU idles)
                                   (4100, 4096, 128), (16384, 512, 256)])
@pytest.mark.parametrize("out_f32", [False, True])
def test_gemm_bias(hip, ref, gemm_epi, M, N, K, out_f32):
    A = rnd(M, K)
    w, W = packed(N, K)
    bias = rnd(N, dtype=torch.float32, seed=3)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if out_f32 else BF16)
    hip.gemm(A, W, out, N=N, K=K, bias=bias, out_f32=out_f32)
    want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias)
    assert rel_err(out.float(), want) < (TOL_F32 if out_f32 else TOL_BF16)


def test_gemm_race_screen(hip, ref):
    """The GEMM kernel keeps the LDS-DMA loads of the next K tile in flight under the MFMAs: repeated launches of a
    many-K-tile problem must be bit-identical (a RAW/WAR race shows up as run-to-run differences) and correct."""
    M, N, K = 3000, 2560, 6912
    A = rnd(M, K)
    w, W = packed(N, K)
    outs = []
    for _ in range(4):
        out = torch.empty(M, N, device="cuda", dtype=BF16)
        hip.gemm(A, W, out, N=N, K=K)
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K)
    assert rel_err(outs[0].float(), want) < TOL_BF16


def test_gemm_transpose_detecting(hip):
    """A = I against an asymmetric W must return W^T rows exactly (catches swapped C/D layouts)."""
    K = N = 256
    A = torch.eye(K, device="cuda", dtype=BF16)
    w = (torch.arange(N * K, device="cuda").reshape(N, K) % 251).to(BF16)
    W = sub("packing").pack_matrix(w, "cuda")
    out = torch.empty(K, N, device="cuda", dtype=BF16)
    hip.gemm(A, W, out, N=N, K=K)
    assert torch.equal(out, w.t().contiguous())


def test_gemm_epilogues(hip, ref, gemm_epi):
    M, N, K = 777, 512, 320
    A = rnd(M, K)
    w, W = packed(N, K)
    bias = rnd(N, dtype=torch.float32, seed=3)
    gate = rnd(N, dtype=torch.float32, seed=4)
    resid = rnd(M, N, seed=5)
    for epi, kw in ((EPI_BIAS_SILU, {}), (EPI_BIAS_GELU, {}), (EPI_RESID_GATE, dict(gate=gate, resid=resid)),
                    (EPI_RESID_GATE, dict(resid=resid)), (EPI_RESID_GATE, dict(gate=gate))):
        out = torch.empty(M, N, device="cuda", dtype=BF16)
        hip.gemm(A, W, out, N=N, K=K, bias=bias, epilogue=epi, **kw)
        want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias, epilogue=epi, **kw)
        assert rel_err(out.float(), want) < TOL_BF16, epi
    # fp32-store test epilogue: the north star's 1e-3 bound, per fused epilogue (same inputs, no output rounding)
    for epi, kw in ((EPI_BIAS_SILU, {}), (EPI_BIAS_GELU, {}), (EPI_RESID_GATE, dict(gate=gate, resid=resid))):
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        hip.gemm(A, W, out, N=N, K=K, bias=bias, epilogue=epi, out_f32=True, **kw)
        want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias, epilogue=epi, **kw)
        assert rel_err(out, want) < TOL_F32, epi
    # in-place residual (C aliases resid), row-sliced views as the DiT uses them
    buf = rnd(M + 58, N, seed=6)
    want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE,
                    gate=gate, resid=buf[:M].clone())
    tail = buf[M:].clone()
    hip.gemm(A, W, buf[:M], N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=buf[:M])
    assert rel_err(buf[:M].float(), want) < TOL_BF16
    assert torch.equal(buf[M:], tail)


def test_gemm_swiglu(hip, ref, gemm_epi):
    packing = sub("packing")
    M, K, Hd = 515, 256, 768
    A = rnd(M, K)
    wg, wi = rnd(Hd, K, scale=1 / 16, seed=7), rnd(Hd, K, scale=1 / 16, seed=8)
    W = packing.pack_swiglu(wg, wi, "cuda")
    out = torch.empty(M, Hd, device="cuda", dtype=BF16)
    hip.gemm(A, W, out, N=2 * Hd, K=K, epilogue=EPI_SWIGLU)
    want = torch.nn.functional.silu(A.float() @ wg.float().t()) * (A.float() @ wi.float().t())
    assert rel_err(out.float(), want) < TOL_BF16
    out32 = torch.empty(M, Hd, device="cuda", dtype=torch.float32)                 # fp32-store test epilogue: 1e-3
    hip.gemm(A, W, out32, N=2 * Hd, K=K, epilogue=EPI_SWIGLU, out_f32=True)
    assert rel_err(out32, want) < TOL_F32
    # and the reference double agrees with the closed form (keeps the two test backends honest)
    w2 = ref.gemm(A, W, torch.empty(M, Hd, device="cuda"), N=2 * Hd, K=K, epilogue=EPI_SWIGLU)
    assert rel_err(w2, want) < 1e-5


@pytest.fixture(params=[0, 1], ids=["gemm_8waves", "gemm_w4q"])
def gemm_big(request, hip):
    """The two main loops for the NaDiT's big plain GEMMs (N % 256 == 0, >= 256 tiles): gemm_kernel (eight waves, 16x16x32 MFMAs) and
    gemm_w4q_kernel (persistent workgroups of four waves of 128 x 128, 16x16x32 MFMAs, hand-scheduled; svr_set_option("gemm_w4"))."""
    hip.set_option("gemm_w4", request.param)
    yield request.param
    hip.set_option("gemm_w4", GEMM_W4_DEFAULT)


GEMM_W4_DEFAULT = 1
GEMM_W4R_DEFAULT = 1


@pytest.mark.parametrize("M,N,K", [(16384, 4096, 128), (16300, 4096, 192), (9000, 7680, 320), (70000, 256, 2560), (4100, 4096, 6912)])
def test_gemm_big_tiles_both_main_loops(hip, ref, gemm_big, M, N, K):
    """Every fused epilogue of the big-GEMM path on both main loops (2, 3, 5, 40 and 108 K tiles; ragged last row panel) against
    the fp32 restatement (fp32 store <= 1e-3, bf16 store <= 2.5e-3), repeated launches bit-identical (LDS-DMA two K tiles ahead
    with hand-counted waits: a race shows up as run-to-run differences)."""
    packing = sub("packing")
    A = rnd(M, K)
    w, W = packed(N, K)
    bias, gate = rnd(N, dtype=torch.float32, seed=3), rnd(N, dtype=torch.float32, seed=4)
    resid = rnd(M, N, seed=5)
    hid = rnd(M, N, seed=6, dtype=torch.float32)
    cases = [dict(bias=bias), dict(bias=bias, epilogue=EPI_BIAS_GELU), dict(bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=resid),
             dict(bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=hid, out_f32=True), dict(out_f32=True)]
    for kw in cases:
        outs = []
        for _ in range(3):
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if kw.get("out_f32") else BF16)
            hip.gemm(A, W, out, N=N, K=K, **kw)
            outs.append(out)
        torch.cuda.synchronize()
        assert not torch.isnan(outs[0].float()).any() and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, **{k: v for k, v in kw.items() if k != "out_f32"})
        assert rel_err(outs[0].float(), want) < (TOL_F32 if kw.get("out_f32") else TOL_BF16), kw.get("epilogue", 0)
        if gemm_big == 1:
            # gemm_w4r_kernel (ABI v7): the same launch with the fragment-ordered weight copy -- weights straight into registers,
            # only the activations through LDS; same MFMAs in the same k order, so the SAME BITS as gemm_w4q_kernel, launch after launch
            # (hand-counted vmcnt waits over 24 loads per K tile: a wrong count shows up as a different result)
            Wf = hip.pack_gemm_frag(W)
            assert Wf is not None
            for rep in range(4):
                out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if kw.get("out_f32") else BF16)
                hip.gemm(A, W, out, N=N, K=K, W_frag=Wf, **kw)
                assert torch.equal(out, outs[0]), ("w4r != w4q", rep, kw.get("epilogue", 0))
    # SwiGLU (interleaved gate | in weights): N = 2 x hidden
    Hd = N // 2
    wg, wi = rnd(Hd, K, scale=1.0 / math.sqrt(K), seed=7), rnd(Hd, K, scale=1.0 / math.sqrt(K), seed=8)
    o = torch.full((M, Hd), float("nan"), device="cuda", dtype=BF16)
    hip.gemm(A, packing.pack_swiglu(wg, wi, "cuda"), o, N=N, K=K, epilogue=EPI_SWIGLU)
    want = torch.nn.functional.silu(A.float() @ wg.float().t()) * (A.float() @ wi.float().t())
    assert not torch.isnan(o.float()).any() and rel_err(o.float(), want) < TOL_BF16
    if gemm_big == 1:
        Wsw = packing.pack_swiglu(wg, wi, "cuda")
        o2 = torch.full((M, Hd), float("nan"), device="cuda", dtype=BF16)
        for rep in range(2):
            o2 = torch.full((M, Hd), float("nan"), device="cuda", dtype=BF16)
            hip.gemm(A, Wsw, o2, N=N, K=K, epilogue=EPI_SWIGLU, W_frag=hip.pack_gemm_frag(Wsw))
            assert torch.equal(o2, o), rep


def test_gemm_pack_frag_layout_is_the_documented_one(hip):
    """svr_gemm_pack_frag (include/seedvr2_hip.h, ABI v7): 16-byte unit (((n / 128) * (K / 32) + k / 32) * 8 + (n % 128) / 16) * 64 + lane
    holds W[(n & ~15) + (lane & 15)][(k & ~31) + (lane >> 4) * 8 .. + 7] -- restated with torch index arithmetic."""
    N, K = 384, 192
    W = torch.arange(N * K, device="cuda", dtype=torch.int32).remainder(32749).to(torch.int16).view(torch.bfloat16).reshape(N, K).contiguous()
    got = hip.pack_gemm_frag(torch.cat([W, W[:128]]).contiguous())      # (N must be a multiple of 256 for the persistent kernel: 512 rows)
    Wp = torch.cat([W, W[:128]])
    Np = Wp.shape[0]
    unit = torch.arange(Np * K // 8, device="cuda")
    lane, u = unit % 64, unit // 64
    J, v = u % 8, u // 8
    ks, p = v % (K // 32), v // (K // 32)
    n = p * 128 + J * 16 + (lane % 16)
    k0 = ks * 32 + (lane // 16) * 8
    want = Wp.view(torch.int16)[n[:, None], k0[:, None] + torch.arange(8, device="cuda")[None, :]].reshape(-1)
    assert torch.equal(got.view(torch.int16), want)


def test_gemm_w4r_option_and_routing(hip):
    """W_frag changes the kernel, not the route: the classifier still says gemm_persistent, svr_set_option("gemm_w4r", 0) sends the
    launch back through gemm_w4q_kernel (bit-identical), and a W_frag too small for the problem is refused."""
    M, N, K = 20000, 2560, 256
    A = rnd(M, K)
    w, W = packed(N, K)
    Wf = hip.pack_gemm_frag(W)
    outs = []
    hip.record_kernel_class = True
    for opt in (1, 0):
        hip.set_option("gemm_w4r", opt)
        try:
            out = torch.empty(M, N, device="cuda", dtype=BF16)
            hip.gemm(A, W, out, N=N, K=K, W_frag=Wf)
            assert hip.last_kernel_class == "gemm_persistent"
            outs.append(out)
        finally:
            hip.set_option("gemm_w4r", GEMM_W4R_DEFAULT)
            hip.record_kernel_class = False
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0].float(), A.float() @ w.float().t()) < TOL_BF16
    with pytest.raises(ValueError):
        hip.gemm(A, W, torch.empty(M, N, device="cuda", dtype=BF16), N=N, K=K, W_frag=Wf[: N * K - 8])
    assert hip.pack_gemm_frag(W[:128]) is None and hip.pack_gemm_frag(W[:, :64].contiguous()) is None


# ------------------------------------------------------------------ implicit-GEMM causal conv
CONV_ROWS_DEFAULT = 8        # the library's default for svr_set_option("conv_rows", ...)
CONV_CASES = [
    # Cin, Cout, k, stride, pad(lo,hi), T, H, W, halo_frames
    (128, 128, (3, 3, 3), (1, 1, 1), (1, 1), 3, 10, 12, 0),
    (128, 128, (3, 3, 3), (1, 1, 1), (1, 1), 2, 9, 7, 2),
    (128, 256, (3, 3, 3), (1, 1, 1), (1, 1), 5, 8, 8, 0),
    (256, 256, (3, 3, 3), (2, 2, 2), (0, 1), 5, 12, 10, 0),
    (256, 256, (3, 3, 3), (2, 2, 2), (0, 1), 4, 12, 10, 1),
    (128, 128, (1, 3, 3), (1, 2, 2), (0, 1), 3, 14, 16, 0),
    (512, 256, (1, 1, 1), (1, 1, 1), (0, 0), 3, 6, 5, 0),
    (512, 32, (3, 3, 3), (1, 1, 1), (1, 1), 2, 6, 7, 0),
    (128, 3, (3, 3, 3), (1, 1, 1), (1, 1), 3, 9, 11, 2),
    # LDS-halo kernel: several 8x32 patches with ragged edges, 2 and 4 channel slices, 1x3x3, N tiles
    (128, 128, (3, 3, 3), (1, 1, 1), (1, 1), 3, 21, 70, 0),
    (256, 128, (3, 3, 3), (1, 1, 1), (1, 1), 2, 17, 33, 2),
    (128, 384, (3, 3, 3), (1, 1, 1), (1, 1), 2, 8, 64, 0),
    (192, 128, (1, 3, 3), (1, 1, 1), (1, 1), 2, 9, 31, 0),
    # thin outputs on the LDS-halo kernel's 32-cout variant (decoder conv_out 128->3, encoder conv_out 512->32)
    (128, 3, (3, 3, 3), (1, 1, 1), (1, 1), 3, 40, 70, 0),
    (512, 32, (3, 3, 3), (1, 1, 1), (1, 1), 2, 19, 45, 2),
]


@pytest.mark.parametrize("conv_impl", [0, pytest.param(1, marks=pytest.mark.variants), pytest.param(-1, marks=pytest.mark.variants), -2],
                         ids=["halo16x32", "generic", "wreg_4rows", "wreg_8rows"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_implicit_gemm(hip, ref, case, conv_impl):
    """conv_impl 0: the library's choice (16x32-voxel LDS-halo kernel where eligible), 1: the generic implicit-GEMM kernel
    everywhere (geometries the halo kernel does not accept run on it anyway).  "wreg_*": the library's
    choice with the fragment-ordered weight copy supplied (weights streamed to registers, not through LDS), with 4
    patch rows per wave and two workgroups per CU, or 8 rows per wave and one wave per SIMD (conv_rows)."""
    hip.set_option("conv_impl", max(conv_impl, 0))
    hip.set_option("conv_rows", 8 if conv_impl == -2 else 4)
    try:
        _conv_case(hip, ref, case, frag=conv_impl < 0)
    finally:
        hip.set_option("conv_impl", 0)
        hip.set_option("conv_rows", CONV_ROWS_DEFAULT)


@pytest.mark.parametrize("case", [CONV_CASES[0], CONV_CASES[1], CONV_CASES[9], CONV_CASES[10], CONV_CASES[3], CONV_CASES[7]],
                         ids=["halo_small", "halo_carried_head", "halo_ragged_patches", "halo_256_128", "generic_stride2", "thin_out"])
@pytest.mark.parametrize("frag", [False, True], ids=["lds_weights", "wreg"])
def test_conv3d_fp32_store_1e3(hip, ref, case, frag):
    """The 1e-3 contract (fp32-store test epilogue, bias + residual fused) for the conv kernels that carry the VAE:
    LDS-halo kernel with LDS-staged and with register-streamed weights, the generic strided kernel, the thin-output one."""
    _conv_case(hip, ref, case, frag=frag, out_f32=True)


def _conv_case(hip, ref, case, frag=False, out_f32=False):
    packing, opsmod = sub("packing"), sub("ops")
    Cin, Cout, k, stride, (plo, phi), T, H, W, hf = case
    kt, kh, kw = k
    x = rnd(T, H, W, Cin)
    halo = rnd(hf, H, W, Cin, seed=9) if hf else None
    w5 = rnd(Cout, Cin, kt, kh, kw, scale=1.0 / math.sqrt(Cin * kt * kh * kw), seed=2)
    Wp = packing.pack_conv3d(w5, "cuda")
    bias = rnd(Cout, dtype=torch.float32, seed=3)
    pt = hf if hf else kt - 1
    To = (T + pt - kt) // stride[0] + 1
    Ho = (H + plo + phi - kh) // stride[1] + 1
    Wo = (W + plo + phi - kw) // stride[2] + 1
    geom = opsmod.Conv3dGeom(T, H, W, Cin, To, Ho, Wo, k, stride, (pt, plo, plo), halo)
    resid = rnd(To, Ho, Wo, Cout, seed=11)
    out = torch.empty(To, Ho, Wo, Cout, device="cuda", dtype=torch.float32 if out_f32 else BF16)
    Wf = hip.pack_conv_frag(Wp, kt, Cin, Cout) if frag and (kh, kw) == (3, 3) else None
    hip.gemm(x, Wp, out, N=Cout, K=Wp.shape[1], bias=bias, conv=geom, epilogue=EPI_RESID_GATE, resid=resid,
             ldc=Cout, ldr=Cout, W_frag=Wf, out_f32=out_f32)
    want = ref.gemm(x, Wp, torch.empty(To, Ho, Wo, Cout, device="cuda"), N=Cout, K=Wp.shape[1], bias=bias,
                    conv=geom, epilogue=EPI_RESID_GATE, resid=resid)
    # independent check of the reference double itself against F.conv3d semantics of the causal conv
    head = halo.float() if hf else x[:1].float().expand(pt, H, W, Cin)
    xin = torch.cat([head, x.float()], 0).permute(3, 0, 1, 2)[None]
    xin = torch.nn.functional.pad(xin, (plo, phi, plo, phi))
    y = torch.nn.functional.conv3d(xin, w5.float(), bias, stride=stride)[0].permute(1, 2, 3, 0) + resid.float()
    assert rel_err(want, y) < 1e-5
    assert rel_err(out.float(), want) < (TOL_F32 if out_f32 else TOL_BF16)


def test_conv3d_halo_kernel_race_screen_and_generic_agreement(hip):
    """The LDS-halo conv keeps loads in flight across barriers: repeated launches must be bit-identical,
    and it must agree with the generic implicit-GEMM kernel (same K order => same fp32 sums up to
    MFMA-shape-dependent rounding)."""
    packing, opsmod = sub("packing"), sub("ops")
    T, H, W, Cin, Cout = 4, 96, 160, 256, 256
    x = rnd(T, H, W, Cin)
    w5 = rnd(Cout, Cin, 3, 3, 3, scale=1.0 / math.sqrt(Cin * 27), seed=2)
    Wp = packing.pack_conv3d(w5, "cuda")
    bias = rnd(Cout, dtype=torch.float32, seed=3)
    geom = opsmod.Conv3dGeom(T, H, W, Cin, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
    outs = []
    for _ in range(3):
        out = torch.empty(T, H, W, Cout, device="cuda", dtype=torch.float32)
        hip.gemm(x, Wp, out, N=Cout, K=Wp.shape[1], bias=bias, conv=geom, ldc=Cout, out_f32=True)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # weights streamed to registers from the fragment-ordered copy: same MFMAs in the same order -> same bits,
    # in both shapes of that kernel (4 rows per wave, two workgroups per CU / 8 rows per wave, one wave per SIMD)
    Wf = hip.pack_conv_frag(Wp, 3, Cin, Cout)
    try:
        for rows in (4, 8):
            hip.set_option("conv_rows", rows)
            for _ in range(4):
                out = torch.full((T, H, W, Cout), float("nan"), device="cuda", dtype=torch.float32)
                hip.gemm(x, Wp, out, N=Cout, K=Wp.shape[1], bias=bias, conv=geom, ldc=Cout, out_f32=True, W_frag=Wf)
                assert torch.equal(out, outs[0]), rows
    finally:
        hip.set_option("conv_rows", CONV_ROWS_DEFAULT)
    hip.set_option("conv_impl", 1)
    try:
        gen = torch.empty(T, H, W, Cout, device="cuda", dtype=torch.float32)
        hip.gemm(x, Wp, gen, N=Cout, K=Wp.shape[1], bias=bias, conv=geom, ldc=Cout, out_f32=True)
    finally:
        hip.set_option("conv_impl", 0)
    assert rel_err(outs[0], gen) < 1e-5


@pytest.mark.parametrize("kt,T,H,W,hf,resid_on", [(3, 3, 21, 70, 0, False), (3, 2, 9, 33, 2, True), (1, 2, 8, 32, 0, False),
                                                   (3, 5, 64, 96, 0, True)])
def test_conv3d_thin_input_fused(hip, ref, kt, T, H, W, hf, resid_on):
    """RGB (Cin 3 -> padded 4) 3x3 conv served by the thin-input variant of the LDS-halo kernel (im2col image built in
    LDS) == the im2col + GEMM route == fp32 F.conv3d; fused GroupNorm statistics == statistics of the stored tensor."""
    packing, opsmod = sub("packing"), sub("ops")
    Cout = 128
    x = rnd(T, H, W, 4)
    x[..., 3] = 0
    halo = rnd(hf, H, W, 4, seed=9) if hf else None
    w5 = rnd(Cout, 3, kt, 3, 3, scale=1.0 / math.sqrt(27 * 3), seed=2)
    Wp = packing.pack_conv3d(w5, "cuda", 4)
    assert Wp.shape == (128, 128) if kt == 3 else Wp.shape[1] % 64 == 0
    if Wp.shape[1] != 128:
        Wp = torch.nn.functional.pad(Wp, (0, 128 - Wp.shape[1]))
    bias = rnd(Cout, dtype=torch.float32, seed=3)
    pt = hf if hf else kt - 1
    To = T + pt - kt + 1
    geom = opsmod.Conv3dGeom(T, H, W, 4, To, H, W, (kt, 3, 3), (1, 1, 1), (pt, 1, 1), halo)
    resid = rnd(To, H, W, Cout, seed=11) if resid_on else None
    epi = EPI_RESID_GATE if resid_on else EPI_BIAS
    out = torch.empty(To, H, W, Cout, device="cuda", dtype=BF16)
    _, stats = hip.gemm(x, Wp, out, N=Cout, K=128, bias=bias, conv=geom, epilogue=epi, resid=resid, ldc=Cout, ldr=Cout,
                        gn_groups=32)
    want = ref.gemm(x, Wp, torch.empty(To, H, W, Cout, device="cuda"), N=Cout, K=128, bias=bias, conv=geom,
                    epilogue=epi, resid=resid)
    assert rel_err(out.float(), want) < TOL_BF16
    # the older route: explicit im2col + plain GEMM
    cols = torch.empty(To * H * W, 128, device="cuda", dtype=BF16)
    hip.im2col_causal(x, cols, geom)
    out2 = torch.empty_like(out)
    hip.gemm(cols, Wp, out2, N=Cout, K=128, M=To * H * W, bias=bias, epilogue=epi, resid=resid, lda=128, ldc=Cout, ldr=Cout)
    assert rel_err(out.float(), out2.float()) < TOL_BF16
    assert stats is not None
    o = out.double().reshape(To, H * W, 32, Cout // 32)
    want_stats = torch.stack([o.sum(dim=(1, 3)), (o * o).sum(dim=(1, 3))], dim=-1)
    assert torch.allclose(stats, want_stats, rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("rz,drop", [(1, False), (2, False), (2, True)])
def test_upscale_pixel_shuffle_epilogue(hip, ref, gemm_epi, rz, drop):
    packing, opsmod = sub("packing"), sub("ops")
    F_, H, W, Cc = 3, 5, 6, 256
    x = rnd(F_ * H * W, Cc)
    w, Wp = packed(4 * rz * Cc, Cc)
    bias = rnd(4 * rz * Cc, dtype=torch.float32, seed=3)
    ps = opsmod.PixelShuffleGeom(F_, H, W, rz, Cc, drop)
    To = F_ * rz - (1 if drop else 0)
    out = torch.full((To, 2 * H, 2 * W, Cc), float("nan"), device="cuda", dtype=BF16)
    hip.gemm(x, Wp, out, N=4 * rz * Cc, K=Cc, M=F_ * H * W, bias=bias, ps=ps)
    want = ref.gemm(x, Wp, torch.empty(To, 2 * H, 2 * W, Cc, device="cuda"), N=4 * rz * Cc, K=Cc, M=F_ * H * W,
                    bias=bias, ps=ps)
    # closed form: "b (x y z c) f h w -> b c (f z) (h x) (w y)"
    y = (x.float() @ w.float().t() + bias).reshape(F_, H, W, 2, 2, rz, Cc)
    y = y.permute(0, 5, 1, 3, 2, 4, 6).reshape(F_ * rz, 2 * H, 2 * W, Cc)
    if drop:
        y = torch.cat([y[:1], y[2:]], 0)
    assert rel_err(want, y) < 1e-5
    assert not torch.isnan(out.float()).any()
    assert rel_err(out.float(), want) < TOL_BF16


def test_gemm_epilogue_paths_bit_identical(hip):
    """The LDS-staged epilogue performs the direct epilogue's arithmetic in the same order: identical bits, for every fused
    epilogue, the fp32 test store, the pixel-shuffle scatter, the generic (strided / 1x1) conv and ragged M."""
    packing, opsmod = sub("packing"), sub("ops")

    def both(fn):
        outs = []
        for mode in (1, 2):
            hip.set_option("gemm_epi", mode)
            try:
                outs.append(fn())
            finally:
                hip.set_option("gemm_epi", 0)
        torch.cuda.synchronize()
        assert outs[0].dtype == outs[1].dtype and not torch.isnan(outs[0].float()).any()
        assert torch.equal(outs[0], outs[1])

    M, N, K = 1531, 768, 192
    A = rnd(M, K)
    _, W = packed(N, K)
    bias, gate, resid = rnd(N, dtype=torch.float32, seed=3), rnd(N, dtype=torch.float32, seed=4), rnd(M, N, seed=5)
    for epi, kw in ((EPI_BIAS, {}), (EPI_BIAS_SILU, {}), (EPI_BIAS_GELU, {}), (EPI_RESID_GATE, dict(gate=gate, resid=resid)),
                    (EPI_RESID_GATE, dict(resid=resid))):
        for f32 in (False, True):
            def run(epi=epi, kw=kw, f32=f32):
                out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if f32 else BF16)
                hip.gemm(A, W, out, N=N, K=K, bias=bias, epilogue=epi, out_f32=f32, **kw)
                return out
            both(run)
    # wide (256 x 256) tiles, many of them, K = 2 tiles: the pixel-shuffle upsampler's shape class
    F_, H, Wd, Cc, rz = 3, 40, 48, 256, 2
    x = rnd(F_ * H * Wd, Cc)
    _, Wp = packed(4 * rz * Cc, Cc)
    b2 = rnd(4 * rz * Cc, dtype=torch.float32, seed=3)
    for drop in (False, True):
        def run_ps(drop=drop):
            ps = opsmod.PixelShuffleGeom(F_, H, Wd, rz, Cc, drop)
            out = torch.full((F_ * rz - (1 if drop else 0), 2 * H, 2 * Wd, Cc), float("nan"), device="cuda", dtype=BF16)
            hip.gemm(x, Wp, out, N=4 * rz * Cc, K=Cc, M=F_ * H * Wd, bias=b2, ps=ps)
            return out
        both(run_ps)
    # SwiGLU
    Hd = 1024
    Wsw = packing.pack_swiglu(rnd(Hd, K, scale=1 / 16, seed=7), rnd(Hd, K, scale=1 / 16, seed=8), "cuda")
    for f32 in (False, True):
        def run_sw(f32=f32):
            out = torch.full((M, Hd), float("nan"), device="cuda", dtype=torch.float32 if f32 else BF16)
            hip.gemm(A, Wsw, out, N=2 * Hd, K=K, epilogue=EPI_SWIGLU, out_f32=f32)
            return out
        both(run_sw)
    # generic implicit-GEMM conv: stride 2 with a residual, and a 1x1x1 shortcut conv
    for Cin, Cout, k, stride, pads in ((256, 256, (3, 3, 3), (2, 2, 2), (0, 1)), (256, 128, (1, 1, 1), (1, 1, 1), (0, 0))):
        T, Hh, Ww = 5, 24, 20
        xin = rnd(T, Hh, Ww, Cin)
        w5 = rnd(Cout, Cin, *k, scale=1.0 / math.sqrt(Cin * k[0] * k[1] * k[2]), seed=2)
        Wc = packing.pack_conv3d(w5, "cuda")
        pt = k[0] - 1
        To = (T + pt - k[0]) // stride[0] + 1
        Ho = (Hh + pads[0] + pads[1] - k[1]) // stride[1] + 1
        Wo = (Ww + pads[0] + pads[1] - k[2]) // stride[2] + 1
        geom = opsmod.Conv3dGeom(T, Hh, Ww, Cin, To, Ho, Wo, k, stride, (pt, pads[0], pads[0]), None)
        rs = rnd(To, Ho, Wo, Cout, seed=11)
        bc = rnd(Cout, dtype=torch.float32, seed=3)

        def run_conv():
            out = torch.full((To, Ho, Wo, Cout), float("nan"), device="cuda", dtype=BF16)
            hip.gemm(xin, Wc, out, N=Cout, K=Wc.shape[1], bias=bc, conv=geom, epilogue=EPI_RESID_GATE, resid=rs,
                     ldc=Cout, ldr=Cout)
            return out
        both(run_conv)


@pytest.mark.parametrize("T,H,W,Cin,Cout,hf,kt,ts", [(3, 9, 11, 128, 128, 0, 3, 1), (2, 16, 20, 256, 256, 2, 3, 1), (4, 1, 5, 64, 128, 0, 3, 1),
                                                     (3, 7, 9, 128, 128, 1, 2, 2), (1, 4, 4, 64, 128, 0, 1, 1)])
@pytest.mark.parametrize("frag", [False, True], ids=["generic_kernel", "subpixel_kernel"])
def test_conv_phase_scatter_subpixel(hip, ref, gemm_epi, frag, T, H, W, Cin, Cout, hf, kt, ts):
    """Sub-pixel convolution launches (svr_gemm_args.phase): (kt, 2, 2)-tap convs over the low-resolution input, each
    scattering into its spatial phase of the 2x grid -- and, with t_stride 2, into every other frame from its first one on --
    with its own border bias == the torch restatement; together they write every voxel of the output exactly once (NaN canary)."""
    packing, opsmod = sub("packing"), sub("ops")
    x = rnd(T, H, W, Cin)
    halo = rnd(hf, H, W, Cin, seed=9) if hf else None
    pt = hf if hf else kt - 1
    To = T + pt - kt + 1
    out = torch.full((To * ts, 2 * H, 2 * W, Cout), float("nan"), device="cuda", dtype=BF16)
    want = torch.full((To * ts, 2 * H, 2 * W, Cout), float("nan"), device="cuda", dtype=torch.float32)
    out_quad = torch.full_like(out, float("nan"))
    for tz in range(ts):                                                           # temporal phase = first frame of the launch
        quad = []
        for ph, (py, px) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            w5 = rnd(Cout, Cin, kt, 2, 2, scale=1.0 / math.sqrt(Cin * 4 * kt), seed=20 + ph + 7 * tz)
            Wp = packing.pack_conv3d(w5, "cuda")
            bias = rnd(Cout, dtype=torch.float32, seed=30 + ph)
            bb = rnd(3, Cout, dtype=torch.float32, seed=40 + ph)
            geom = opsmod.Conv3dGeom(T, H, W, Cin, To, H, W, (kt, 2, 2), (1, 1, 1), (pt, 1 - py, 1 - px), halo)
            kw = dict(N=Cout, K=Wp.shape[1], bias=bias, conv=geom, phase=opsmod.PhaseScatter(py, px, bb, ts))
            # frag: the fragment-ordered weight copy selects the sub-pixel conv kernel (svr_conv_sub.hip; Cin % 64 == 0)
            Wf = hip.pack_conv_frag(Wp, kt, Cin, Cout, taps=(2, 2)) if frag else None
            assert Wf is not None or not frag
            quad.append((py, px, Wp, bias, bb, Wf))
            if ph == 3 and frag:
                # the same four phases as ONE launch (svr_phase_scatter.quad: phase fastest in the tile order): same bits
                q0 = quad[0]
                g0 = opsmod.Conv3dGeom(T, H, W, Cin, To, H, W, (kt, 2, 2), (1, 1, 1), (pt, 1, 1), halo)
                assert hip._quad_ok(x, q0[2], out_quad[tz:], Cout, q0[2].shape[1], g0, opsmod.PhaseScatter(0, 0, q0[4], ts, quad=quad), False)
                hip.gemm(x, q0[2], out_quad[tz:], N=Cout, K=q0[2].shape[1], bias=q0[3], conv=g0, W_frag=q0[5],
                         phase=opsmod.PhaseScatter(0, 0, q0[4], ts, quad=quad))
            hip.gemm(x, Wp, out[tz:], W_frag=Wf, **kw)
            ref.gemm(x, Wp, want[tz:], **kw)
            # independent check of the restatement on this phase: F.conv3d of the padded input + per-voxel bias
            head = halo.float() if hf else x[:1].float().expand(pt, H, W, Cin)
            xin = torch.nn.functional.pad(torch.cat([head, x.float()], 0).permute(3, 0, 1, 2)[None], (1 - px, px, 1 - py, py))
            y = torch.nn.functional.conv3d(xin, w5.float())[0].permute(1, 2, 3, 0)
            b = bias.expand(To, H, W, Cout).clone()
            rb, cb = (H - 1 if py else 0), (W - 1 if px else 0)
            b[:, rb] = bb[0]
            b[:, :, cb] = bb[1]
            b[:, rb, cb] = bb[2]
            assert rel_err(want[tz::ts, py::2, px::2], y + b) < 1e-5
    assert not torch.isnan(out.float()).any() and not torch.isnan(want).any()
    assert rel_err(out.float(), want) < TOL_BF16
    if frag:
        assert torch.equal(out_quad, out)


@pytest.mark.parametrize("T,H,W,Cin,Cout,hf,kt,ts", [(3, 9, 11, 128, 128, 0, 3, 1), (2, 40, 70, 256, 256, 2, 3, 1), (3, 17, 33, 128, 256, 1, 2, 2),
                                                     (1, 4, 4, 64, 128, 0, 1, 1)])
def test_conv_subpixel_fused_groupnorm_statistics(hip, T, H, W, Cin, Cout, hf, kt, ts):
    """The sub-pixel conv kernel's fused GroupNorm statistics: the phase launches of an upsampled tensor share one partial
    buffer [frame][phase][block][group] (ops.gemm(gn_shared=...)); reduced, it equals svr_groupnorm_stats of the stored tensor
    (both sum the bf16 values that were written; fixed orders, so repeated runs are bit-identical), frames offset by frame0."""
    packing, opsmod = sub("packing"), sub("ops")
    G = 32
    x = rnd(T, H, W, Cin)
    halo = rnd(hf, H, W, Cin, seed=9) if hf else None
    pt = hf if hf else kt - 1
    To = T + pt - kt + 1
    lead = 2                                                                       # frames in front of the launches' first frame
    runs = []
    for _ in range(2):
        out = torch.zeros((lead + To * ts, 2 * H, 2 * W, Cout), device="cuda", dtype=BF16)
        shared = {"frames": out.shape[0]}
        # the lead frames get their four phases from a launch of their own (as a head launch of the engine would); then one
        # launch group per temporal phase: (first output frame, input, halo, causal pad, frames, frame stride)
        groups = [(0, x[:1].expand(lead, H, W, Cin).contiguous(), None, kt - 1, lead, 1)]
        groups += [(lead + tz, x, halo, pt, To, ts) for tz in range(ts)]
        for base, xin, hl, p_t, to_n, stride in groups:
            for ph, (py, px) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                w5 = rnd(Cout, Cin, kt, 2, 2, scale=1.0 / math.sqrt(Cin * 4 * kt), seed=20 + ph + base)
                Wp = packing.pack_conv3d(w5, "cuda")
                geom = opsmod.Conv3dGeom(xin.shape[0], H, W, Cin, to_n, H, W, (kt, 2, 2), (1, 1, 1), (p_t, 1 - py, 1 - px), hl)
                shared["frame0"] = base
                hip.gemm(xin, Wp, out[base:], N=Cout, K=Wp.shape[1], bias=rnd(Cout, dtype=torch.float32, seed=30 + ph), conv=geom,
                         phase=opsmod.PhaseScatter(py, px, rnd(3, Cout, dtype=torch.float32, seed=40 + ph), stride),
                         W_frag=hip.pack_conv_frag(Wp, kt, Cin, Cout, taps=(2, 2)), gn_groups=G, gn_shared=shared)
        stats = hip.gn_shared_stats(shared)
        assert stats is not None and tuple(stats.shape) == (out.shape[0], G, 2)
        # the same launches in their quad form (one launch per group): same output, same statistics, bit for bit
        out_q = torch.zeros_like(out)
        shared_q = {"frames": out.shape[0]}
        for base, xin, hl, p_t, to_n, stride in groups:
            quad = []
            for ph, (py, px) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                Wp = packing.pack_conv3d(rnd(Cout, Cin, kt, 2, 2, scale=1.0 / math.sqrt(Cin * 4 * kt), seed=20 + ph + base), "cuda")
                quad.append((py, px, Wp, rnd(Cout, dtype=torch.float32, seed=30 + ph), rnd(3, Cout, dtype=torch.float32, seed=40 + ph),
                             hip.pack_conv_frag(Wp, kt, Cin, Cout, taps=(2, 2))))
            geom = opsmod.Conv3dGeom(xin.shape[0], H, W, Cin, to_n, H, W, (kt, 2, 2), (1, 1, 1), (p_t, 1, 1), hl)
            shared_q["frame0"] = base
            hip.gemm(xin, quad[0][2], out_q[base:], N=Cout, K=quad[0][2].shape[1], bias=quad[0][3], conv=geom, W_frag=quad[0][5],
                     phase=opsmod.PhaseScatter(0, 0, quad[0][4], stride, quad=quad), gn_groups=G, gn_shared=shared_q)
        assert torch.equal(out_q, out) and torch.equal(hip.gn_shared_stats(shared_q), stats)
        want = torch.empty(out.shape[0], G, 2, dtype=torch.float64, device="cuda")
        hip.groupnorm_stats(out, want, G)
        assert rel_err(stats[..., 0], want[..., 0]) < 1e-5 and rel_err(stats[..., 1], want[..., 1]) < 1e-6
        runs.append(stats)
    assert torch.equal(runs[0], runs[1])


@pytest.mark.parametrize("Cin,Cout,kt,T,H,W,hf", [(128, 3, 3, 3, 40, 70, 0), (512, 32, 3, 2, 19, 45, 2), (128, 3, 1, 2, 9, 33, 0), (192, 16, 3, 4, 8, 32, 0)])
def test_conv_thin_output_kernel(hip, ref, Cin, Cout, kt, T, H, W, hf):
    """Thin-output convs (N <= 32: decoder conv_out 128 -> 3, encoder conv_out 512 -> 32) on the step-interval kernel
    (svr_conv_thinout.hip) == the torch restatement, == the generic implicit-GEMM kernel up to MFMA-order rounding,
    plain bias epilogue (production) and residual epilogue; repeated launches bit-identical (double-buffered LDS-DMA)."""
    packing, opsmod = sub("packing"), sub("ops")
    x = rnd(T, H, W, Cin)
    halo = rnd(hf, H, W, Cin, seed=9) if hf else None
    w5 = rnd(Cout, Cin, kt, 3, 3, scale=1.0 / math.sqrt(Cin * 9 * kt), seed=2)
    Wp = packing.pack_conv3d(w5, "cuda")
    bias = rnd(Cout, dtype=torch.float32, seed=3)
    pt = hf if hf else kt - 1
    To = T + pt - kt + 1
    geom = opsmod.Conv3dGeom(T, H, W, Cin, To, H, W, (kt, 3, 3), (1, 1, 1), (pt, 1, 1), halo)
    resid = rnd(To, H, W, Cout, seed=11)
    for kw in (dict(), dict(epilogue=EPI_RESID_GATE, resid=resid, ldr=Cout)):
        kw = dict(N=Cout, K=Wp.shape[1], bias=bias, conv=geom, ldc=Cout, **kw)
        outs = []
        for _ in range(3):
            out = torch.full((To, H, W, Cout), float("nan"), device="cuda", dtype=BF16)
            hip.gemm(x, Wp, out, **kw)
            outs.append(out)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        want = ref.gemm(x, Wp, torch.empty(To, H, W, Cout, device="cuda"), **kw)
        assert not torch.isnan(outs[0].float()).any() and rel_err(outs[0].float(), want) < TOL_BF16
        hip.set_option("conv_impl", 1)                      # the generic implicit-GEMM kernel on the same launch
        try:
            old = torch.empty(To, H, W, Cout, device="cuda", dtype=BF16)
            hip.gemm(x, Wp, old, **kw)
        finally:
            hip.set_option("conv_impl", 0)
        assert rel_err(outs[0].float(), old.float()) < 2e-3


@pytest.mark.parametrize("kt,ts,H,W", [(2, 2, 40, 70), (3, 1, 33, 64), (1, 1, 16, 32)])
def test_conv_subpixel_kernel_race_screen_and_generic_agreement(hip, kt, ts, H, W):
    """The sub-pixel conv kernel keeps LDS-DMA, weight loads and fragment reads in flight across barriers with hand-counted
    waits: repeated launches over many tiles and A steps must be bit-identical, and must agree with the generic
    implicit-GEMM kernel on the same launch (conv_sub 0) up to the rounding of differently shaped MFMA sums."""
    packing, opsmod = sub("packing"), sub("ops")
    T, Cin, Cout = 3, 256, 256
    x = rnd(T, H, W, Cin)
    halo = rnd(kt - 1, H, W, Cin, seed=9) if kt > 1 else None
    w5 = rnd(Cout, Cin, kt, 2, 2, scale=1.0 / math.sqrt(Cin * 4 * kt), seed=2)
    Wp = packing.pack_conv3d(w5, "cuda")
    Wf = hip.pack_conv_frag(Wp, kt, Cin, Cout, taps=(2, 2))
    bias, bb = rnd(Cout, dtype=torch.float32, seed=3), rnd(3, Cout, dtype=torch.float32, seed=4)
    geom = opsmod.Conv3dGeom(T, H, W, Cin, T, H, W, (kt, 2, 2), (1, 1, 1), (kt - 1, 1, 0), halo)
    kw = dict(N=Cout, K=Wp.shape[1], bias=bias, conv=geom, phase=opsmod.PhaseScatter(0, 1, bb, ts), W_frag=Wf)
    outs = []
    for _ in range(4):
        out = torch.zeros(T * ts, 2 * H, 2 * W, Cout, device="cuda", dtype=BF16)
        hip.gemm(x, Wp, out, **kw)
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    hip.set_option("conv_sub", 0)
    try:
        gen = torch.zeros_like(outs[0])
        hip.gemm(x, Wp, gen, **kw)
    finally:
        hip.set_option("conv_sub", 1)
    assert torch.equal(gen[:, 1::2], outs[0][:, 1::2]) and torch.equal(gen[:, :, 0::2], outs[0][:, :, 0::2])     # untouched phases stay zero
    assert rel_err(outs[0].float(), gen.float()) < 2e-3
    # and without the phase scatter: a plain (kt, 2, 2) conv into a dense tensor
    kw2 = dict(N=Cout, K=Wp.shape[1], bias=bias, conv=geom, W_frag=Wf, ldc=Cout)
    d1 = torch.full((T, H, W, Cout), float("nan"), device="cuda", dtype=BF16)
    hip.gemm(x, Wp, d1, **kw2)
    hip.set_option("conv_sub", 0)
    try:
        d0 = torch.full((T, H, W, Cout), float("nan"), device="cuda", dtype=BF16)
        hip.gemm(x, Wp, d0, **kw2)
    finally:
        hip.set_option("conv_sub", 1)
    assert not torch.isnan(d1.float()).any() and rel_err(d1.float(), d0.float()) < 2e-3


def test_vae_subpixel_upsampler_matches_two_step_on_gpu(hip):
    """The engine's sub-pixel upsampler against the reference's two steps on the device (bf16 storage both): they differ by
    one rounding of the merged weights / of the upsampled intermediate, far below the parity budget of the decoder."""
    config, weights, vae_mod = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg, device="cuda")
    z = (torch.randn(5, 12, 10, cfg.latent_channels, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.5).to(BF16)
    a = vae_mod.VideoVAEEngine(cfg, sd, hip, merge_upsamplers=False, merge_causal_head=False).decode(z).float()
    eng = vae_mod.VideoVAEEngine(cfg, sd, hip)
    assert all(up is None or up.merged is not None for _, up in eng.dec_up)
    b = eng.decode(z).float()
    e = rel_err(b, a)
    print(f"sub-pixel vs two-step upsamplers, decode rel-err {e:.3e}")
    assert a.shape == b.shape and e < 2e-2          # (two bf16 paths, each ~1e-2 from the fp32 reference: tests/test_gpu_parity.py holds the real bound)
    for per_slice in (1, 2, 3):                                                # temporal slicing stays bit-exact
        assert torch.equal(eng.decode(z, latents_per_slice=per_slice).float(), b), per_slice
    one = eng.decode(z[:1]).float()                                            # a single latent frame: the head pattern only
    assert rel_err(one, vae_mod.VideoVAEEngine(cfg, sd, hip, merge_upsamplers=False, merge_causal_head=False).decode(z[:1]).float()) < 2e-2


def test_vae_causal_head_two_term_sum_matches_three_taps_on_gpu(hip):
    """Frame 0 of every clip with the three temporal taps on the replicated first frame folded into a (hi, lo) pair of bf16
    weights (vae.py:_conv_causal_head): same function to 2^-17 per weight.  One layer: frame 0 differs from the three-tap
    launch by isolated bf16 roundings only, frames 1.. and their statistics are bit-identical; whole network: the two bf16
    paths sit at the usual bf16 noise floor of a random-weight VAE from each other (tests/test_gpu_parity.py holds the real
    bound against the reference); slicing stays bit-exact; a single image takes the head launch alone."""
    config, weights, vae_mod = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(2)
    three = vae_mod.VideoVAEEngine(cfg, sd, hip, merge_causal_head=False)
    two = vae_mod.VideoVAEEngine(cfg, sd, hip)
    for pick in (lambda e: e.dec_up[0][0][0].conv1, lambda e: e.dec_up[3][0][0].conv1, lambda e: e.dec_conv_out):
        c2, c3 = pick(two), pick(three)
        assert c2.head is not None and c3.head is None and (c2.head.w_frag is not None) == (c2.w_frag is not None)
        for T in (1, 2, 5):
            x = torch.randn(T, 40, 64, c2.cin, device="cuda", generator=gen).to(BF16)
            gn = c2.cout % 128 == 0
            r2, r3 = (e._conv(c, x, {"__last_slice__": True}, True, gn=gn) for e, c in ((two, c2), (three, c3)))
            o2, o3 = (r2[0], r3[0]) if gn else (r2, r3)
            assert torch.equal(o2[1:], o3[1:]), (c2.name, T)
            assert rel_err(o2[:1].float(), o3[:1].float()) < 1e-3, (c2.name, T)
            if gn and r2[1] is not None and r3[1] is not None:
                assert torch.equal(r2[1][1:], r3[1][1:]) and rel_err(r2[1][:1], r3[1][:1]) < 1e-4
    z = (torch.randn(3, 12, 10, cfg.latent_channels, device="cuda", generator=gen) * 0.5).to(BF16)
    x = (torch.rand(3, 9, 64, 96, device="cuda", generator=gen) * 2 - 1).to(BF16)
    d3, d2 = three.decode(z).float(), two.decode(z).float()
    e3, e2 = three.encode(x).float(), two.encode(x).float()
    print(f"two-term head vs three taps: decode rel-err {rel_err(d2, d3):.3e}, encode rel-err {rel_err(e2, e3):.3e}")
    assert d2.shape == d3.shape and rel_err(d2, d3) < 2e-2 and rel_err(e2, e3) < 2e-2
    assert torch.equal(two.decode(z, latents_per_slice=1).float(), d2) and torch.equal(two.encode(x, frames_per_slice=4).float(), e2)
    assert rel_err(two.decode(z[:1]).float(), three.decode(z[:1]).float()) < 2e-2
    assert rel_err(two.encode(x[:, :1]).float(), three.encode(x[:, :1]).float()) < 2e-2


@pytest.fixture(params=[pytest.param(0, marks=pytest.mark.variants), pytest.param(4, marks=pytest.mark.variants), 8],
                ids=["lds_weights", "wreg_4rows", "wreg_8rows"])
def conv_variant(request, hip):
    """The three shapes of the second LDS-halo conv kernel: weights through the LDS ring (no fragment-ordered copy), weights
    streamed to registers with 4 or with 8 patch rows per wave.  -> rows per wave (0: no fragment-ordered copy)."""
    hip.set_option("conv_rows", request.param or CONV_ROWS_DEFAULT)
    yield request.param
    hip.set_option("conv_rows", CONV_ROWS_DEFAULT)


@pytest.mark.parametrize("Cin,Cout,resid", [(128, 128, True), (256, 128, False), (128, 256, True), (128, 512, False)])
def test_conv_fused_groupnorm_stats(hip, ref, conv_variant, Cin, Cout, resid):
    """GroupNorm (sum, sumsq) fused into the LDS-halo conv epilogue == statistics of the tensor it stored
    (same bf16 values; fp64 reductions in a fixed order), bit-reproducible, ragged patches masked."""
    packing, opsmod = sub("packing"), sub("ops")
    T, H, W = 3, 37, 70
    x = rnd(T, H, W, Cin)
    w5 = rnd(Cout, Cin, 3, 3, 3, scale=1.0 / math.sqrt(Cin * 27), seed=2)
    Wp = packing.pack_conv3d(w5, "cuda")
    bias = rnd(Cout, dtype=torch.float32, seed=3)
    geom = opsmod.Conv3dGeom(T, H, W, Cin, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
    res = rnd(T, H, W, Cout, seed=11) if resid else None
    kw = dict(N=Cout, K=Wp.shape[1], bias=bias, conv=geom, ldc=Cout, ldr=Cout,
              epilogue=EPI_RESID_GATE if resid else EPI_BIAS, resid=res,
              W_frag=hip.pack_conv_frag(Wp, 3, Cin, Cout) if conv_variant else None)
    out = torch.empty(T, H, W, Cout, device="cuda", dtype=BF16)
    got, stats = hip.gemm(x, Wp, out, gn_groups=32, **kw)
    assert got is out and stats is not None and stats.shape == (T, 32, 2)
    plain = torch.empty_like(out)
    hip.gemm(x, Wp, plain, **kw)
    assert torch.equal(out, plain)                         # the fused statistics do not change the output
    want = ref.groupnorm_stats(out, torch.empty(T, 32, 2, device="cuda", dtype=torch.float64), 32)
    assert torch.allclose(stats, want, rtol=1e-6, atol=1e-6)
    _, again = hip.gemm(x, Wp, torch.empty_like(out), gn_groups=32, **kw)
    assert torch.equal(stats, again)
    # a geometry the halo kernel does not take: no fused statistics, the caller falls back
    g2 = opsmod.Conv3dGeom(T, H, W, Cin, T, (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1, (1, 3, 3), (1, 2, 2), (0, 0, 0), None)
    w2 = packing.pack_conv3d(rnd(Cout, Cin, 1, 3, 3, scale=0.05, seed=4), "cuda")
    o2 = torch.empty(T, g2.Ho, g2.Wo, Cout, device="cuda", dtype=BF16)
    _, none = hip.gemm(x, w2, o2, N=Cout, K=w2.shape[1], bias=bias, conv=g2, ldc=Cout, gn_groups=32)
    assert none is None


# ------------------------------------------------------------------ DiT side kernels
@pytest.mark.parametrize("rows,dim", [(1000, 2560), (58, 2560), (333, 256), (7, 3072)])
def test_rmsnorm_mod(hip, ref, rows, dim):
    x = rnd(rows, dim, scale=2.0)
    w, sc, sh = (rnd(dim, dtype=torch.float32, seed=s) for s in (1, 2, 3))
    for kw in (dict(), dict(scale=sc, shift=sh), dict(w=w, scale=sc, shift=sh)):
        out = torch.empty(rows, dim, device="cuda", dtype=BF16)
        hip.rmsnorm_mod(x, out, 1e-5, **kw)
        want = ref.rmsnorm_mod(x, torch.empty(rows, dim, device="cuda"), 1e-5, **kw)
        assert rel_err(out.float(), want) < TOL_BF16


def test_ada_combine(hip, ref):
    dim, nv = 2560, 13
    emb, params = rnd(dim * 6), rnd(nv, dim, seed=2)
    slots = torch.tensor([0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 1], dtype=torch.int32, device="cuda")
    out = torch.empty(nv, dim, device="cuda", dtype=torch.float32)
    hip.ada_combine(emb, params, slots, out)
    want = ref.ada_combine(emb, params, slots, torch.empty(nv, dim, device="cuda"))
    assert torch.allclose(out, want, atol=1e-6)


def _rope_tables(n_pos):
    freqs = sub("weights").rope_freqs_lang(42)
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * freqs[None, :]
    return ang.cos().cuda().contiguous(), ang.sin().cuda().contiguous()


def test_qknorm_rope(hip, ref):
    rows, heads = 500, 20
    qkv = rnd(rows, 3 * heads * 128)
    g = torch.Generator().manual_seed(0)
    pos = torch.stack([torch.randint(0, 4, (rows,), generator=g), torch.randint(0, 15, (rows,), generator=g),
                       torch.randint(0, 27, (rows,), generator=g)], -1).to(torch.int16).cuda()
    cos, sin = _rope_tables(100)
    wq, wk = rnd(128, dtype=torch.float32, seed=1) + 1, rnd(128, dtype=torch.float32, seed=2) + 1
    want = ref.qknorm_rope(qkv.float().clone(), heads, pos, 58, cos, sin, wq, wk, 1e-5)
    got = qkv.clone()
    hip.qknorm_rope(got, heads, pos, 58, cos, sin, wq, wk, 1e-5)
    assert torch.equal(got[:, 2 * heads * 128:], qkv[:, 2 * heads * 128:])          # V untouched
    assert rel_err(got.float(), want) < TOL_BF16


def _attn_case(lens, heads, D, n_rows, seed=0):
    g = torch.Generator().manual_seed(seed)
    seq_rows, out_rows, cu = [], [], [0]
    o = 0
    for L in lens:
        seq_rows.append(torch.randint(0, n_rows, (L,), generator=g))
        out_rows.append(torch.arange(o, o + L))
        o += L
        cu.append(o)
    to = lambda t: torch.cat(t).to(torch.int32).cuda()
    return to(seq_rows), to(out_rows), torch.tensor(cu, dtype=torch.int32).cuda(), o


@pytest.fixture(params=[0, pytest.param(1, marks=pytest.mark.variants)], ids=["attn_win", "attn_gen1"])
def attn_impl(request, hip):
    """Both window-attention kernels stay parity-green: 0 = second-generation kernel (svr_attn_win.hip) wherever it
    applies (head_dim 128, windows <= 2048 rows), 1 = the first kernel everywhere."""
    hip.set_option("attn_impl", request.param)
    yield request.param
    hip.set_option("attn_impl", 0)


@pytest.mark.parametrize("lens,heads,D", [([135, 64, 1, 200, 129, 1273], 3, 128), ([314], 20, 128),
                                          ([100, 33, 257], 1, 512),
                                          # tile-boundary lengths (64-key tiles, 128-query tiles), 1- and 2-row windows
                                          ([64, 128, 192, 63, 65, 127, 129, 2, 1, 256], 2, 128),
                                          # (window, head) pairs not a multiple of the 8 XCDs; longest window the LDS row table holds
                                          ([2048, 77, 1215, 640], 5, 128),
                                          # one row more than the table: served by the first kernel whatever attn_impl says
                                          ([2049, 300], 1, 128)])
def test_attn_varlen(hip, ref, attn_impl, lens, heads, D, request):
    n_rows = 1500
    qkv = rnd(n_rows, 3 * heads * D)
    seq_rows, out_rows, cu, total = _attn_case(lens, heads, D, n_rows)
    scale = 1.0 / math.sqrt(D)
    out = torch.full((total, heads * D), float("nan"), device="cuda", dtype=BF16)     # every row must be written
    hip.attn_varlen(qkv, out, seq_rows, out_rows, cu, max(lens), heads, D, scale)
    want = ref.attn_varlen(qkv, torch.zeros(total, heads * D, device="cuda"), seq_rows, out_rows, cu, max(lens),
                           heads, D, scale)
    assert rel_err(out.float(), want) < 4e-3       # P is rounded to bf16 before PV (as flash kernels do)
    # race screen: the double-buffered LDS pipeline must be deterministic
    for _ in range(3):
        again = torch.zeros_like(out)
        hip.attn_varlen(qkv, again, seq_rows, out_rows, cu, max(lens), heads, D, scale)
        assert torch.equal(again, out)
    if attn_impl == 0 and D == 128 and "variants" in (request.config.getoption("-m") or ""):   # the kernel's build variants (A/B knob)
        for variant in (1, 3, 4):   # 4 waves + s_setprio, 8 waves + s_setprio, 4 waves (default: 8 waves)
            hip.set_option("attn_variant", variant)
            try:
                other = torch.full_like(out, float("nan"))
                hip.attn_varlen(qkv, other, seq_rows, out_rows, cu, max(lens), heads, D, scale)
            finally:
                hip.set_option("attn_variant", 0)
            assert torch.equal(other, out), variant     # same MFMAs in the same order per query row -> same bits


def test_attn_varlen_scattered_output_rows(hip, ref, attn_impl):
    """Window-ordered gather AND scatter through distinct index vectors (text rows go to a scratch tail), output rows
    that no window owns stay untouched."""
    heads, D, n_rows = 4, 128, 3000
    lens = [700, 411, 1273, 90]
    qkv = rnd(n_rows, 3 * heads * D, seed=3)
    g = torch.Generator().manual_seed(11)
    perm = torch.randperm(n_rows, generator=g)
    total = sum(lens)
    seq_rows = perm[:total].to(torch.int32).cuda()
    out_rows = torch.randperm(total + 500, generator=g)[:total].to(torch.int32).cuda()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32).cuda()
    out = torch.full((total + 500, heads * D), 7.0, device="cuda", dtype=BF16)
    hip.attn_varlen(qkv, out, seq_rows, out_rows, cu, max(lens), heads, D, 1.0 / math.sqrt(D))
    want = ref.attn_varlen(qkv, torch.full((total + 500, heads * D), 7.0, device="cuda"), seq_rows, out_rows, cu,
                           max(lens), heads, D, 1.0 / math.sqrt(D))
    assert rel_err(out.float(), want) < 4e-3
    untouched = torch.ones(total + 500, dtype=torch.bool, device="cuda")
    untouched[out_rows.long()] = False
    assert bool((out[untouched] == 7.0).all())


def test_attn_varlen_spiky_scores(hip, ref, attn_impl):
    """Forces the online-softmax rescale: one key dominates late in the sequence (and, for the second-generation kernel,
    the wave-uniform skip of the rescale when no running max moved is exercised by the tiles after it)."""
    heads, D, L = 2, 128, 400
    qkv = rnd(L, 3 * heads * D, scale=0.5)
    qkv[300, heads * D:2 * heads * D] = qkv[5, :heads * D] * 8          # k[300] aligned with q[5]
    rows = torch.arange(L, dtype=torch.int32, device="cuda")
    cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    out = torch.zeros(L, heads * D, device="cuda", dtype=BF16)
    hip.attn_varlen(qkv, out, rows, rows, cu, L, heads, D, 1.0 / math.sqrt(D))
    want = ref.attn_varlen(qkv, torch.zeros(L, heads * D, device="cuda"), rows, rows, cu, L, heads, D, 1.0 / math.sqrt(D))
    assert rel_err(out.float(), want) < 4e-3
    assert rel_err(out[5].float(), want[5]) < 4e-3                      # the row whose max jumps at key 300


def test_attn_varlen_constant_scores_skip_rescale(hip, ref, attn_impl):
    """All keys identical -> every score of a row is equal, the running max never moves after the first tile, so the
    second-generation kernel takes its no-rescale branch on every later tile; the output must equal v (uniform
    attention over identical values would hide an error, so V differs per key: out = mean(v))."""
    heads, D, L = 1, 128, 640
    qkv = rnd(L, 3 * heads * D, seed=5)
    qkv[:, D:2 * D] = qkv[0, D:2 * D]
    rows = torch.arange(L, dtype=torch.int32, device="cuda")
    cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    out = torch.zeros(L, D, device="cuda", dtype=BF16)
    hip.attn_varlen(qkv, out, rows, rows, cu, L, heads, D, 1.0 / math.sqrt(D))
    want = qkv[:, 2 * D:].float().mean(0, keepdim=True).expand(L, D)
    assert rel_err(out.float(), want) < 4e-3


def test_rows_mean_patchify_unpatchify(hip, ref):
    src = rnd(7 * 58, 2560)
    dst = torch.empty(58, 2560, device="cuda", dtype=BF16)
    hip.rows_mean(src, dst, 7, 58)
    assert rel_err(dst.float(), ref.rows_mean(src, torch.empty(58, 2560, device="cuda"), 7, 58)) < TOL_BF16
    vid = rnd(3, 8, 12, 33)
    out = torch.empty(3 * 4 * 6, 192, device="cuda", dtype=BF16)
    hip.patchify(vid, out)
    assert torch.equal(out, ref.patchify(vid, torch.empty(3 * 4 * 6, 192, device="cuda", dtype=BF16)))
    pred, x_t = rnd(3 * 4 * 6, 64), rnd(3, 8, 12, 16, seed=4)
    for xt in (x_t, None):
        o = torch.empty(3, 8, 12, 16, device="cuda", dtype=BF16)
        hip.unpatchify_euler(pred, xt, o)
        assert rel_err(o.float(), ref.unpatchify_euler(pred, xt, torch.empty(3, 8, 12, 16, device="cuda"))) < TOL_BF16


# ------------------------------------------------------------------ VAE side kernels
@pytest.mark.parametrize("C", [128, 256, 512])
def test_groupnorm(hip, ref, C):
    T, H, W = 3, 37, 41
    x = rnd(T, H, W, C, scale=1.5) + 0.7
    gamma, beta = rnd(C, dtype=torch.float32, seed=1) + 1, rnd(C, dtype=torch.float32, seed=2)
    stats = torch.empty(T, 32, 2, device="cuda", dtype=torch.float64)
    hip.groupnorm_stats(x, stats, 32)
    want_stats = ref.groupnorm_stats(x, torch.empty(T, 32, 2, device="cuda", dtype=torch.float64), 32)
    assert torch.allclose(stats, want_stats, rtol=1e-5)
    again = torch.empty_like(stats)
    hip.groupnorm_stats(x, again, 32)
    assert torch.equal(stats, again)                       # fixed-order reduction: bit-reproducible
    one = torch.empty(1, 32, 2, device="cuda", dtype=torch.float64)
    hip.groupnorm_stats(x[1:2].contiguous(), one, 32)
    assert torch.equal(one[0], stats[1])                   # and independent of the frame's position in the slice
    for silu in (True, False):
        out = torch.empty_like(x)
        hip.groupnorm_apply(x, out, stats, gamma, beta, 32, 1e-6, silu)
        want = ref.groupnorm_apply(x, torch.empty(T, H, W, C, device="cuda"), want_stats, gamma, beta, 32, 1e-6, silu)
        gn = torch.nn.functional.group_norm(x.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-6).permute(0, 2, 3, 1)
        assert rel_err(want, torch.nn.functional.silu(gn) if silu else gn) < 1e-4
        assert rel_err(out.float(), want) < TOL_BF16


@pytest.mark.parametrize("Cin,kpad,hf", [(4, 128, 0), (4, 128, 2), (16, 448, 0)])
def test_im2col_causal(hip, ref, Cin, kpad, hf):
    opsmod = sub("ops")
    T, H, W = 3, 9, 10
    x = rnd(T, H, W, Cin)
    halo = rnd(hf, H, W, Cin, seed=5) if hf else None
    geom = opsmod.Conv3dGeom(T, H, W, Cin, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), halo)
    out = torch.empty(T * H * W, kpad, device="cuda", dtype=BF16)
    hip.im2col_causal(x, out, geom)
    assert torch.equal(out, ref.im2col_causal(x, torch.empty(T * H * W, kpad, device="cuda", dtype=BF16), geom))


def test_blend_and_affine(hip, ref):
    T, h, w, C, H, W = 2, 5, 6, 32, 9, 11
    tile = rnd(T, h, w, C)
    wy, wx = torch.rand(h, device="cuda"), torch.rand(w, device="cuda")
    acc, cnt = torch.zeros(T, H, W, C, device="cuda"), torch.zeros(H, W, device="cuda")
    acc2, cnt2 = acc.clone(), cnt.clone()
    for (y0, x0) in ((0, 0), (3, 4), (4, 5)):
        hip.blend_accumulate(tile, acc, cnt, wy, wx, y0, x0)
        ref.blend_accumulate(tile, acc2, cnt2, wy, wx, y0, x0)
    assert torch.allclose(acc, acc2, atol=1e-5) and torch.allclose(cnt, cnt2, atol=1e-6)
    out = torch.empty(T, H, W, 16, device="cuda", dtype=BF16)
    hip.blend_finalize(acc, cnt, out, 0.9152, 0.1)
    want = ref.blend_finalize(acc2, cnt2, torch.empty(T, H, W, 16, device="cuda"), 0.9152, 0.1)
    assert rel_err(out.float(), want) < TOL_BF16
    inp = rnd(50, 32)
    o = torch.empty(50, 16, device="cuda", dtype=BF16)
    hip.affine_slice(inp, o, 1 / 0.9152, -0.05)
    assert rel_err(o.float(), ref.affine_slice(inp, torch.empty(50, 16, device="cuda"), 1 / 0.9152, -0.05)) < TOL_BF16


@pytest.mark.parametrize("rows,cols", [(64, 64), (100, 4416), (33, 16384), (7, 260), (9, 20000), (5, 65536)])
def test_softmax_rows(hip, rows, cols):
    g = torch.Generator(device="cuda").manual_seed(rows + cols)
    S = torch.randn(rows, cols, device="cuda", generator=g) * 30.0
    P = hip.empty(rows, cols)
    hip.softmax_rows(S, P, 0.044)
    want = torch.softmax(S * 0.044, dim=-1)
    assert torch.isfinite(P.float()).all()
    assert (P.float() - want).abs().max() <= 2 ** -8 * want.max() + 1e-6         # bf16 rounding of the output
    assert (P.float().sum(-1) - 1).abs().max() < 2e-2


@pytest.mark.parametrize("T,H,W", [(2, 16, 24), (1, 128, 160)], ids=["one_block", "two_row_blocks_20480_tokens"])
def test_vae_attention_gemm_path_matches_fused_kernel(hip, T, H, W):
    """Mid-block attention run as QK^T GEMM -> softmax -> PV GEMM (in blocks of 16384 query rows: the second case has 20480 tokens
    per frame, as the untiled frames of BASELINE config 2 have 65536) vs the fused variable-length kernel and fp32 torch."""
    from conftest import sub
    vae_mod, weights, config = sub("vae"), sub("weights"), sub("config")
    cfg = config.VAEConfig(block_out_channels=(128, 128, 128, 128))
    sd = weights.synth_vae_state_dict(cfg, seed=3)
    eng = vae_mod.VideoVAEEngine(cfg, sd, hip)
    ab = eng.enc_mid[1]
    x = (torch.randn(T, H, W, 128, device="cuda") * 0.7).bfloat16()
    h2f = sub("ops").h16_to_float                          # (the block's output is a trunk tensor: h16 by default)
    got = h2f(eng._attention(ab, x))
    eng.attn_as_gemm = False
    fused = h2f(eng._attention(ab, x))
    # fp32 torch restatement of the block (GroupNorm -> q,k,v -> softmax(q k^T / sqrt(C)) v -> out proj + residual)
    name = "encoder.mid_block.attentions.0"
    w = {k: sd[f"{name}.{k}"].float().cuda() for k in
         ("group_norm.weight", "group_norm.bias", "to_q.weight", "to_q.bias", "to_k.weight", "to_k.bias",
          "to_v.weight", "to_v.bias", "to_out.0.weight", "to_out.0.bias")}
    xf = x.float()
    y = torch.nn.functional.group_norm(xf.permute(0, 3, 1, 2), 32, w["group_norm.weight"], w["group_norm.bias"], 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(T, -1, 128)
    q, k, v = (y @ w[f"to_{c}.weight"].T + w[f"to_{c}.bias"] for c in "qkv")
    o = torch.softmax(q @ k.transpose(1, 2) / 128 ** 0.5, -1) @ v
    want = (o @ w["to_out.0.weight"].T + w["to_out.0.bias"]).reshape(xf.shape) + xf
    for name_, t in (("gemm", got), ("fused", fused)):
        err = (t - want).abs()
        assert err.max() < 0.08 and err.mean() < 6e-3, (name_, float(err.max()), float(err.mean()))
    assert (got - fused).abs().max() < 0.08


@pytest.mark.parametrize("H,W", [(256, 256), (270, 480)], ids=["cfg2_65536_tokens", "untiled_4k_129600_tokens"])
def test_vae_attention_at_config2_shape_512_channels_65536_tokens(hip, H, W):
    """BASELINE config 2's untiled 2048x2048 frames put 256 x 256 = 65536 tokens of 512 channels through the mid-block attention
    (attn_video_vae.py:615-665: one head, d = 512, softmax rows of 65536 columns): the engine runs it as four blocks of 16384 query
    rows of Q K^T GEMM (fp32 scores) -> svr_softmax_rows -> P V GEMM (vae.py::_attention).  Checked at exactly that shape against a
    blocked fp32 torch restatement on the device, and against the fused d = 512 kernel.
    Round 6: the same check on an UNTILED 4K frame (270 x 480 = 129 600 tokens) -- above 65 536 tokens _attention routes to
    the fused d = 512 kernel whatever attn_as_gemm says; no test had run it there."""
    from conftest import sub
    vae_mod, weights, config = sub("vae"), sub("weights"), sub("config")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg, seed=3)
    eng = vae_mod.VideoVAEEngine(cfg, sd, hip)
    ab = eng.dec_mid[1]
    C = 512
    n = H * W
    g = torch.Generator(device="cuda").manual_seed(11)
    x = (torch.randn(1, H, W, C, device="cuda", generator=g) * 0.7).bfloat16()
    assert eng.attn_as_gemm
    h2f = sub("ops").h16_to_float                          # (the block's output is a trunk tensor: h16 by default)
    got = h2f(eng._attention(ab, x))
    name = "decoder.mid_block.attentions.0"
    w = {k: sd[f"{name}.{k}"].float().cuda() for k in
         ("group_norm.weight", "group_norm.bias", "to_q.weight", "to_q.bias", "to_k.weight", "to_k.bias",
          "to_v.weight", "to_v.bias", "to_out.0.weight", "to_out.0.bias")}
    xf = x.float()
    y = torch.nn.functional.group_norm(xf.permute(0, 3, 1, 2), 32, w["group_norm.weight"], w["group_norm.bias"], 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(-1, C)
    q, k, v = (y @ w[f"to_{c}.weight"].T + w[f"to_{c}.bias"] for c in "qkv")
    o = torch.empty_like(q)
    blk = 8192 if n <= 65536 else 4050                                 # 8192 x 65536 (4050 x 129600) fp32 scores = 2 GiB per block
    for r0 in range(0, q.shape[0], blk):
        o[r0:r0 + blk] = torch.softmax(q[r0:r0 + blk] @ k.T / C ** 0.5, -1) @ v
    want = (o @ w["to_out.0.weight"].T + w["to_out.0.bias"]).reshape(xf.shape) + xf
    err = (got - want).abs()
    e = float((got - want).norm() / want.norm())
    print(f"VAE mid-block attention, 512 channels x {n} tokens ({'GEMM-softmax-GEMM' if n <= 65536 else 'fused d = 512 kernel'}): "
          f"rel-err {e:.3e}, max abs {float(err.max()):.3e}")
    assert e < 4e-3 and err.max() < 0.08
    if n > 65536:
        return
    eng.attn_as_gemm = False
    fused = h2f(eng._attention(ab, x))
    ef = float((fused - want).norm() / want.norm())
    print(f"  fused d = 512 kernel at the same shape: rel-err {ef:.3e}")
    assert ef < 4e-3 and (got - fused).abs().max() < 0.08


@pytest.mark.parametrize("T,H,W,Cin,N,hf,kt", [(3, 40, 70, 128, 3, 0, 3), (2, 17, 33, 128, 3, 2, 3), (1, 16, 32, 256, 16, 0, 3), (2, 5, 9, 128, 3, 0, 1),
                                               (1, 33, 65, 128, 3, 0, 2)])
@pytest.mark.parametrize("out_f32", [False, True], ids=["bf16_store", "fp32_store"])
def test_conv_thin_output_4_cout_kernel(hip, ref, T, H, W, Cin, N, hf, kt, out_f32):
    """conv_thinout4_kernel (N <= 4 couts on v_mfma_f32_16x16x32_bf16, four patch rows per wave, resident weights, frames written back
    through LDS -- decoder conv_out) against the torch conv and against the 32-cout kernel it replaces on these launches: ragged patches
    (odd row starts for the 2-byte stores), carried temporal halo, the two-tap causal-head launch (kt = 2), 1e-3 with an fp32 store; the
    16-cout case takes the 32-cout kernel either way."""
    packing, opsmod = sub("packing"), sub("ops")
    x = rnd(T, H, W, Cin)
    halo = rnd(hf, H, W, Cin, seed=9) if hf else None
    w5 = rnd(N, Cin, kt, 3, 3, scale=1.0 / math.sqrt(Cin * kt * 9), seed=2)
    Wp = packing.pack_conv3d(w5, "cuda")
    bias = rnd(N, dtype=torch.float32, seed=3)
    pt = hf if hf else kt - 1
    To = T + pt - kt + 1
    geom = opsmod.Conv3dGeom(T, H, W, Cin, To, H, W, (kt, 3, 3), (1, 1, 1), (pt, 1, 1), halo)
    kw = dict(N=N, K=Wp.shape[1], bias=bias, conv=geom, ldc=N)
    want = ref.gemm(x, Wp, torch.empty(To, H, W, N, device="cuda"), **kw)
    outs = []
    for new in (1, 0):
        hip.set_option("conv_thinout4", new)
        try:
            out = torch.full((To, H, W, N), float("nan"), device="cuda", dtype=torch.float32 if out_f32 else BF16)
            hip.gemm(x, Wp, out, out_f32=out_f32, **kw)
            again = torch.full_like(out, float("nan"))
            hip.gemm(x, Wp, again, out_f32=out_f32, **kw)
        finally:
            hip.set_option("conv_thinout4", 1)
        assert not torch.isnan(out.float()).any() and torch.equal(out, again)          # every voxel written, reproducible
        assert rel_err(out.float(), want) < (1e-3 if out_f32 else TOL_BF16), new
        outs.append(out.float())
    assert rel_err(outs[0], outs[1]) < (2e-4 if out_f32 else 4e-3)


def test_mfma_calibrate_reports_a_plausible_rate(hip):
    """svr_mfma_calibrate (ABI v9, measurement aid): a bare v_mfma_f32_32x32x16_bf16 loop timed with events on the launch stream must
    land between the slowest sustained rate seen on the pool and the nominal dense bf16 peak -- bench.py divides the dominant kernel's
    rate by it (roofline.frac_of_power_limited), so a broken FLOP count or a loop the compiler removed must fail here."""
    rate = hip.mfma_calibrate(seconds=0.1)
    print(f"bare-MFMA rate of this device right now: {rate:.0f} TFLOP/s")
    assert 1200.0 < rate < 2500.0, rate
    assert abs(hip.mfma_calibrate(seconds=0.1) - rate) < 0.08 * rate          # (repeatable within the box's clock noise)
