"""-m gpu: the wide residual trunk -- fp32 (ABI v5) or h16 (ABI v6: an IEEE half holding x * 2^-6, ops.H16) outputs / residuals in
the conv and GEMM epilogues, fp32 / h16 inputs of the normalisation kernels -- against the torch restatement of the same ops
(tests/ops_reference.py, which reads and writes the same storage formats) on identical inputs.

Contract: the arithmetic is unchanged (bf16 MFMA operands, fp32 accumulate); what changes is that a value on the skip path is
not rounded to bf16 when it is stored.  So an fp32 store must meet the fp32-store bound 1e-3 (measured ~3e-4: MFMA summation
order), fused GroupNorm statistics are those of the fp32 values that were stored, and a bf16 store after an fp32 residual
meets the bf16-store bound 2.5e-3.  An h16 store rounds once to 11 significant bits (2.8e-4 rms): it meets the fp32-store bound
1e-3 as well (measured ~4e-4), and its fused statistics are those of the h16 values that were stored."""
import math

import pytest
import torch

from conftest import sub, rel_err
from ops_reference import TorchOps, EPI_BIAS, EPI_RESID_GATE, H16, H16_SCALE, _ld

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
TOL_BF16, TOL_F32 = 2.5e-3, 1e-3
WIDE = (F32, H16)


@pytest.fixture(scope="module")
def hip():
    return sub("ops").HipOps("cuda:0")


@pytest.fixture(scope="module")
def ref():
    return TorchOps("cuda:0", act_dtype=torch.float32)


def rnd(*shape, scale=1.0, seed=0, dtype=BF16):
    g = torch.Generator(device="cuda").manual_seed(seed + sum(shape))
    v = torch.randn(*shape, generator=g, device="cuda") * scale
    return (v * H16_SCALE).to(H16) if dtype == H16 else v.to(dtype)      # (h16: the stored half is x * 2^-6)


@pytest.mark.parametrize("frag", [True, pytest.param(False, marks=pytest.mark.variants)], ids=["wreg_8rows", "lds_weights"])
@pytest.mark.parametrize("out_dt,res_dt", [(F32, None), (F32, F32), (F32, BF16), (BF16, F32), (H16, None), (H16, H16), (H16, BF16), (BF16, H16),
                                           (H16, F32)],
                         ids=["f32_out", "f32_out_f32_resid", "f32_out_bf16_resid", "bf16_out_f32_resid", "h16_out", "h16_out_h16_resid",
                              "h16_out_bf16_resid", "bf16_out_h16_resid", "h16_out_f32_resid_runtime_body"])
@pytest.mark.parametrize("Cin,Cout,T,H,W,hf", [(128, 128, 3, 21, 70, 0), (256, 128, 2, 17, 33, 2)])
def test_conv_halo_wide_trunk_epilogues(hip, ref, frag, out_dt, res_dt, Cin, Cout, T, H, W, hf):
    """conv_halo2_kernel's compile-time option sets with fp32 output and / or fp32 residual, with and without the fused
    GroupNorm statistics: output vs the torch conv, statistics == those of the tensor that was stored, launches bit-reproducible."""
    packing, opsmod = sub("packing"), sub("ops")
    x = rnd(T, H, W, Cin)
    halo = rnd(hf, H, W, Cin, seed=9) if hf else None
    w5 = rnd(Cout, Cin, 3, 3, 3, scale=1.0 / math.sqrt(Cin * 27), seed=2)
    Wp = packing.pack_conv3d(w5, "cuda")
    bias = rnd(Cout, dtype=F32, seed=3)
    pt = hf if hf else 2
    To = T + pt - 2
    geom = opsmod.Conv3dGeom(T, H, W, Cin, To, H, W, (3, 3, 3), (1, 1, 1), (pt, 1, 1), halo)
    resid = rnd(To, H, W, Cout, seed=11, dtype=res_dt) if res_dt is not None else None
    kw = dict(N=Cout, K=Wp.shape[1], bias=bias, conv=geom, ldc=Cout, ldr=Cout, resid=resid,
              epilogue=EPI_RESID_GATE if resid is not None else EPI_BIAS, W_frag=hip.pack_conv_frag(Wp, 3, Cin, Cout) if frag else None)
    want = ref.gemm(x, Wp, torch.empty(To, H, W, Cout, device="cuda"), **{k: v for k, v in kw.items() if k != "W_frag"})
    out = torch.full((To, H, W, Cout), float("nan"), device="cuda", dtype=out_dt)
    got, stats = hip.gemm(x, Wp, out, gn_groups=32, out_f32=out_dt in WIDE, **kw)
    assert got is out and stats is not None and not torch.isnan(out.float()).any()
    assert rel_err(_ld(out), want) < (TOL_F32 if out_dt in WIDE else TOL_BF16)
    plain = torch.empty_like(out)
    hip.gemm(x, Wp, plain, out_f32=out_dt in WIDE, **kw)
    assert torch.equal(plain, out)                                   # the fused statistics do not change the output
    ws = ref.groupnorm_stats(out, torch.empty(To, 32, 2, device="cuda", dtype=torch.float64), 32)
    assert rel_err(stats[..., 0], ws[..., 0]) < 1e-5 and rel_err(stats[..., 1], ws[..., 1]) < 1e-6
    _, again = hip.gemm(x, Wp, torch.empty_like(out), gn_groups=32, out_f32=out_dt in WIDE, **kw)
    assert torch.equal(again, stats)


@pytest.mark.parametrize("wide", WIDE, ids=["fp32", "h16"])
def test_conv_halo_wide_trunk_in_place_residual(hip, ref, wide):
    """out aliases the fp32 residual (how a block could update the trunk in place): each thread reads its 8 values before it
    writes them, so the result equals the out-of-place launch."""
    packing, opsmod = sub("packing"), sub("ops")
    T, H, W, C = 2, 19, 45, 128
    x = rnd(T, H, W, C)
    Wp = packing.pack_conv3d(rnd(C, C, 3, 3, 3, scale=1.0 / math.sqrt(C * 27), seed=2), "cuda")
    geom = opsmod.Conv3dGeom(T, H, W, C, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
    trunk = rnd(T, H, W, C, seed=5, dtype=wide)
    kw = dict(N=C, K=Wp.shape[1], bias=rnd(C, dtype=F32, seed=3), conv=geom, ldc=C, ldr=C, epilogue=EPI_RESID_GATE,
              W_frag=hip.pack_conv_frag(Wp, 3, C, C), out_f32=True)
    sep = torch.empty_like(trunk)
    hip.gemm(x, Wp, sep, resid=trunk, **kw)
    inplace = trunk.clone()
    hip.gemm(x, Wp, inplace, resid=inplace, **kw)
    assert torch.equal(inplace, sep)


@pytest.mark.parametrize("wide", WIDE, ids=["fp32", "h16"])
@pytest.mark.parametrize("kt,ts,hf", [(3, 1, 0), (2, 2, 1)])
def test_conv_subpixel_wide_output_and_statistics(hip, ref, kt, ts, hf, wide):
    """The sub-pixel upsampler kernel writing the upsampled tensor in fp32 / h16 (an upsampler whose output stays on the wide
    trunk), all four phases, with the shared fused statistics."""
    packing, opsmod = sub("packing"), sub("ops")
    T, H, W, Cin, Cout, G = 3, 17, 33, 128, 128, 32
    x = rnd(T, H, W, Cin)
    halo = rnd(hf, H, W, Cin, seed=9) if hf else None
    pt = hf if hf else kt - 1
    To = T + pt - kt + 1
    out = torch.full((To * ts, 2 * H, 2 * W, Cout), float("nan"), device="cuda", dtype=wide)
    want = torch.zeros(out.shape, device="cuda")
    shared = {"frames": out.shape[0]}
    for tz in range(ts):
        for ph, (py, px) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            Wp = packing.pack_conv3d(rnd(Cout, Cin, kt, 2, 2, scale=1.0 / math.sqrt(Cin * 4 * kt), seed=20 + ph + 4 * tz), "cuda")
            geom = opsmod.Conv3dGeom(T, H, W, Cin, To, H, W, (kt, 2, 2), (1, 1, 1), (pt, 1 - py, 1 - px), halo)
            kw = dict(N=Cout, K=Wp.shape[1], bias=rnd(Cout, dtype=F32, seed=30 + ph), conv=geom,
                      phase=opsmod.PhaseScatter(py, px, rnd(3, Cout, dtype=F32, seed=40 + ph), ts))
            shared["frame0"] = tz
            hip.gemm(x, Wp, out[tz:], W_frag=hip.pack_conv_frag(Wp, kt, Cin, Cout, taps=(2, 2)), gn_groups=G, gn_shared=shared,
                     out_f32=True, **kw)
            ref.gemm(x, Wp, want[tz:], **kw)
    assert not torch.isnan(out).any() and rel_err(_ld(out), want) < TOL_F32
    stats = hip.gn_shared_stats(shared)
    ws = ref.groupnorm_stats(out, torch.empty(out.shape[0], G, 2, device="cuda", dtype=torch.float64), G)
    # (per-thread fp32 partial sums of fp32 values, then fp64: compare as vectors, like the bf16 test of this kernel)
    assert stats is not None and rel_err(stats[..., 0], ws[..., 0]) < 1e-5 and rel_err(stats[..., 1], ws[..., 1]) < 1e-6


@pytest.mark.parametrize("epi", [1, 2], ids=["epi_direct", "epi_lds"])
@pytest.mark.parametrize("M,N,K", [(777, 512, 320), (3000, 2560, 2560), (58, 2560, 6912)])
def test_gemm_wide_residual_stream(hip, ref, epi, M, N, K):
    """gate + fp32 residual -> fp32 output, in place (the NaDiT residual stream: mmsr_block.py:108-126), both epilogue paths of
    the GEMM kernel; and fp32 residual -> bf16 output (the 7B family's last block)."""
    packing = sub("packing")
    hip.set_option("gemm_epi", epi)
    try:
        A = rnd(M, K)
        W = packing.pack_matrix(rnd(N, K, scale=1.0 / math.sqrt(K), seed=1), "cuda")
        bias, gate = rnd(N, dtype=F32, seed=3), rnd(N, dtype=F32, seed=4)
        hid = rnd(M + 58, N, seed=5, dtype=F32)
        tail = hid[M:].clone()
        want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate,
                        resid=hid[:M].clone())
        narrow = torch.empty(M, N, device="cuda", dtype=BF16)
        hip.gemm(A, W, narrow, N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=hid[:M])
        assert rel_err(narrow.float(), want) < TOL_BF16
        hip.gemm(A, W, hid[:M], N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=hid[:M], out_f32=True)
        assert rel_err(hid[:M], want) < TOL_F32 and torch.equal(hid[M:], tail)
    finally:
        hip.set_option("gemm_epi", 0)


@pytest.mark.parametrize("epi", [1, 2], ids=["epi_direct", "epi_lds"])
@pytest.mark.parametrize("M,N,K,gated", [(777, 512, 512, True), (4099, 512, 512, False), (300, 128, 4096, False)])
def test_gemm_h16_trunk_epilogues(hip, ref, epi, M, N, K, gated):
    """The GEMM kernel's h16 forms (the VAE's mid-block attention output projection + residual on the trunk, the decoder's conv_in
    through im2col, 1x1 convs): bias -> h16; gate * (acc + bias) + h16 residual -> h16, in place; h16 residual -> bf16; bf16
    residual -> h16.  Both epilogue paths of the kernel."""
    packing = sub("packing")
    hip.set_option("gemm_epi", epi)
    try:
        A = rnd(M, K)
        W = packing.pack_matrix(rnd(N, K, scale=1.0 / math.sqrt(K), seed=1), "cuda")
        bias = rnd(N, dtype=F32, seed=3)
        gate = rnd(N, dtype=F32, seed=4) if gated else None
        out = torch.full((M, N), float("nan"), device="cuda", dtype=H16)
        hip.gemm(A, W, out, N=N, K=K, bias=bias, out_f32=True)
        assert rel_err(_ld(out), ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias)) < TOL_F32
        for res_dt, out_dt in ((H16, H16), (H16, BF16), (BF16, H16)):
            res = rnd(M, N, seed=5, dtype=res_dt)
            want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=res)
            o = torch.full((M, N), float("nan"), device="cuda", dtype=out_dt)
            hip.gemm(A, W, o, N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=res, out_f32=out_dt in WIDE)
            assert rel_err(_ld(o), want) < (TOL_F32 if out_dt in WIDE else TOL_BF16), (res_dt, out_dt)
            if res_dt == out_dt:                                       # in place on the trunk
                hip.gemm(A, W, res, N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=res, out_f32=True)
                assert torch.equal(res, o)
    finally:
        hip.set_option("gemm_epi", 0)


def test_generic_conv_h16_output(hip, ref):
    """The generic implicit-GEMM kernel (stride-2 downsampler, 1x1x1 shortcut) storing a trunk tensor in h16."""
    packing, opsmod = sub("packing"), sub("ops")
    for Cin, Cout, k, stride, pads in ((128, 128, (3, 3, 3), (2, 2, 2), (0, 1)), (256, 128, (1, 1, 1), (1, 1, 1), (0, 0))):
        T, Hh, Ww = 5, 24, 20
        x = rnd(T, Hh, Ww, Cin)
        Wc = packing.pack_conv3d(rnd(Cout, Cin, *k, scale=1.0 / math.sqrt(Cin * k[0] * k[1] * k[2]), seed=2), "cuda")
        pt = k[0] - 1
        To, Ho, Wo = (T + pt - k[0]) // stride[0] + 1, (Hh + pads[0] + pads[1] - k[1]) // stride[1] + 1, (Ww + pads[0] + pads[1] - k[2]) // stride[2] + 1
        geom = opsmod.Conv3dGeom(T, Hh, Ww, Cin, To, Ho, Wo, k, stride, (pt, pads[0], pads[0]), None)
        kw = dict(N=Cout, K=Wc.shape[1], bias=rnd(Cout, dtype=F32, seed=3), conv=geom, ldc=Cout)
        out = torch.full((To, Ho, Wo, Cout), float("nan"), device="cuda", dtype=H16)
        hip.gemm(x, Wc, out, out_f32=True, **kw)
        assert rel_err(_ld(out), ref.gemm(x, Wc, torch.empty(To, Ho, Wo, Cout, device="cuda"), **kw)) < TOL_F32


@pytest.mark.parametrize("rows,dim", [(1000, 2560), (58, 2560), (7, 3072), (333, 256)])
def test_rmsnorm_mod_fp32_input(hip, ref, rows, dim):
    x = rnd(rows, dim, scale=2.0, dtype=F32)
    w, sc, sh = (rnd(dim, dtype=F32, seed=s) for s in (1, 2, 3))
    for kw in (dict(), dict(scale=sc, shift=sh), dict(w=w, scale=sc, shift=sh)):
        out = torch.empty(rows, dim, device="cuda", dtype=BF16)
        hip.rmsnorm_mod(x, out, 1e-5, **kw)
        assert rel_err(out.float(), ref.rmsnorm_mod(x, torch.empty(rows, dim, device="cuda"), 1e-5, **kw)) < TOL_BF16
    # a bf16-representable fp32 input gives the bf16 kernel's bits
    xb = rnd(rows, dim, scale=2.0)
    a, b = torch.empty(rows, dim, device="cuda", dtype=BF16), torch.empty(rows, dim, device="cuda", dtype=BF16)
    hip.rmsnorm_mod(xb, a, 1e-5, scale=sc, shift=sh)
    hip.rmsnorm_mod(xb.float(), b, 1e-5, scale=sc, shift=sh)
    assert torch.equal(a, b)


@pytest.mark.parametrize("C", [128, 256, 512])
def test_groupnorm_h16_input(hip, ref, C):
    """GroupNorm statistics and apply (+SiLU) reading an h16 trunk tensor: the statistics are those of the values the halves
    stand for (x * 2^6), the apply kernel folds the 2^6 into its per-channel factor."""
    T, H, W = 3, 37, 41
    g = torch.Generator(device="cuda").manual_seed(C)
    x = ((torch.randn(T, H, W, C, generator=g, device="cuda") * 1.5 + 0.7) * H16_SCALE).to(H16)
    gamma, beta = rnd(C, dtype=F32, seed=1) + 1, rnd(C, dtype=F32, seed=2)
    stats = torch.empty(T, 32, 2, device="cuda", dtype=torch.float64)
    hip.groupnorm_stats(x, stats, 32)
    want_stats = ref.groupnorm_stats(x, torch.empty(T, 32, 2, device="cuda", dtype=torch.float64), 32)
    assert torch.allclose(stats, want_stats, rtol=1e-5)
    assert torch.allclose(stats, ref.groupnorm_stats(_ld(x), torch.empty_like(stats), 32), rtol=1e-5)      # == the fp32 view's
    for silu in (True, False):
        out = torch.empty(T, H, W, C, device="cuda", dtype=BF16)
        hip.groupnorm_apply(x, out, stats, gamma, beta, 32, 1e-6, silu)
        want = ref.groupnorm_apply(x, torch.empty(T, H, W, C, device="cuda"), want_stats, gamma, beta, 32, 1e-6, silu)
        assert rel_err(out.float(), want) < TOL_BF16
    # the h16 kernels on an exactly representable tensor == the fp32 kernels on its fp32 view, bit for bit
    sf = torch.empty_like(stats)
    hip.groupnorm_stats(_ld(x), sf, 32)
    assert torch.equal(stats, sf)
    a, b = torch.empty(T, H, W, C, device="cuda", dtype=BF16), torch.empty(T, H, W, C, device="cuda", dtype=BF16)
    hip.groupnorm_apply(x, a, stats, gamma, beta, 32, 1e-6, True)
    hip.groupnorm_apply(_ld(x), b, stats, gamma, beta, 32, 1e-6, True)
    assert rel_err(a.float(), b.float()) < 2e-4 and (a != b).float().mean() < 0.02    # (x * (a * 64) vs (x * 64) * a: last-bit flips)


@pytest.mark.parametrize("C", [128, 256, 512])
def test_groupnorm_fp32_input(hip, ref, C):
    T, H, W = 3, 37, 41
    x = rnd(T, H, W, C, scale=1.5, dtype=F32) + 0.7
    gamma, beta = rnd(C, dtype=F32, seed=1) + 1, rnd(C, dtype=F32, seed=2)
    stats = torch.empty(T, 32, 2, device="cuda", dtype=torch.float64)
    hip.groupnorm_stats(x, stats, 32)
    want_stats = ref.groupnorm_stats(x, torch.empty(T, 32, 2, device="cuda", dtype=torch.float64), 32)
    assert torch.allclose(stats, want_stats, rtol=1e-5)
    for silu in (True, False):
        out = torch.empty(T, H, W, C, device="cuda", dtype=BF16)
        hip.groupnorm_apply(x, out, stats, gamma, beta, 32, 1e-6, silu)
        want = ref.groupnorm_apply(x, torch.empty(T, H, W, C, device="cuda"), want_stats, gamma, beta, 32, 1e-6, silu)
        assert rel_err(out.float(), want) < TOL_BF16
    xb = rnd(T, H, W, C, scale=1.5)                               # bf16-representable fp32 input == the bf16 kernel, bit for bit
    sb, sf = torch.empty_like(stats), torch.empty_like(stats)
    hip.groupnorm_stats(xb, sb, 32)
    hip.groupnorm_stats(xb.float(), sf, 32)
    assert torch.equal(sb, sf)
    a, b = torch.empty_like(xb), torch.empty_like(xb)
    hip.groupnorm_apply(xb, a, sb, gamma, beta, 32, 1e-6, True)
    hip.groupnorm_apply(xb.float(), b, sb, gamma, beta, 32, 1e-6, True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("wide", WIDE, ids=["fp32", "h16"])
def test_conv_thin_input_wide_output(hip, ref, wide):
    """encoder conv_in (RGB padded to 4 channels) storing the first trunk tensor in fp32 / h16, fused statistics included."""
    packing, opsmod = sub("packing"), sub("ops")
    T, H, W, Cout = 3, 21, 70, 128
    x = rnd(T, H, W, 4)
    x[..., 3] = 0
    Wp = packing.pack_conv3d(rnd(Cout, 3, 3, 3, 3, scale=1.0 / math.sqrt(81), seed=2), "cuda", 4)
    geom = opsmod.Conv3dGeom(T, H, W, 4, T, H, W, (3, 3, 3), (1, 1, 1), (2, 1, 1), None)
    kw = dict(N=Cout, K=Wp.shape[1], bias=rnd(Cout, dtype=F32, seed=3), conv=geom, ldc=Cout)
    out = torch.full((T, H, W, Cout), float("nan"), device="cuda", dtype=wide)
    _, stats = hip.gemm(x, Wp, out, gn_groups=32, out_f32=True, **kw)
    want = ref.gemm(x, Wp, torch.empty(T, H, W, Cout, device="cuda"), **kw)
    assert rel_err(_ld(out), want) < TOL_F32 and stats is not None
    ws = ref.groupnorm_stats(out, torch.empty(T, 32, 2, device="cuda", dtype=torch.float64), 32)
    assert rel_err(stats[..., 0], ws[..., 0]) < 1e-5 and rel_err(stats[..., 1], ws[..., 1]) < 1e-6


def test_vae_engine_storage_regimes_agree_with_their_cpu_emulation(hip):
    """The engine with and without the wide trunk on the HIP path vs the same host code over the torch double of the C ABI in
    the same storage regime (bf16 activations, fp32 trunk): the two regimes are different functions (fewer roundings), each
    must match its own emulation to the rounding noise of that regime."""
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAEConfig(block_out_channels=(128, 256, 256, 512))
    sd = weights.synth_vae_state_dict(cfg, seed=7)
    z = rnd(2, 6, 8, 16, seed=3)
    errs = {}
    for regime in (dict(trunk_store="h16", branch_store="h16"), dict(trunk_store="fp32", branch_store="bf16"),
                   dict(trunk_store="bf16", branch_store="bf16")):
        name = regime["trunk_store"] + "/" + regime["branch_store"]
        eng = vae.VideoVAEEngine(cfg, sd, hip, **regime)
        got = eng.decode(z).float().cpu()
        emu = vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), **regime).decode(z.cpu()).float()
        errs[name] = rel_err(got, emu)
        print(f"decoder, trunk/branch {name}: HIP vs CPU emulation of the same storage regime rel-err {errs[name]:.3e}")
        # two runs of one regime differ by the regime's own rounding noise (each is ~0.7e-2 / ~1e-2 from the exact function)
        assert got.shape == emu.shape and errs[name] < 2e-2, name
        x = rnd(3, 5, 48, 64, seed=4)                      # the encoder in the same regime (thin conv_in, strided convs, shortcuts)
        e_enc = rel_err(eng.encode(x).float().cpu(),
                        vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), **regime).encode(x.cpu()).float())
        print(f"encoder, trunk/branch {name}: rel-err {e_enc:.3e}")
        assert e_enc < 2e-2, name
    assert vae.VideoVAEEngine(cfg, sd, hip).trunk_store == "h16"           # the default regime
    assert errs["h16/h16"] < errs["bf16/bf16"] and errs["fp32/bf16"] < errs["bf16/bf16"]   # fewer roundings -> less noise


# ------------------------------------------------------------------ the NaDiT's 2-byte residual stream (round 5)
@pytest.mark.parametrize("rows,dim", [(1000, 2560), (58, 2560), (7, 3072)])
def test_rmsnorm_mod_h16_input(hip, ref, rows, dim):
    """RMSNorm + modulation reading the stream in h16 (the stored half is x * 2^-6; eps applies to the values the halves stand for)."""
    xv = rnd(rows, dim, scale=2.0, dtype=F32)
    x = (xv * H16_SCALE).to(H16)
    w, sc, sh = (rnd(dim, dtype=F32, seed=s) for s in (1, 2, 3))
    for kw in (dict(), dict(scale=sc, shift=sh), dict(w=w, scale=sc, shift=sh)):
        out = torch.empty(rows, dim, device="cuda", dtype=BF16)
        hip.rmsnorm_mod(x, out, 1e-5, **kw)
        assert rel_err(out.float(), ref.rmsnorm_mod(x, torch.empty(rows, dim, device="cuda"), 1e-5, **kw)) < TOL_BF16
        same = torch.empty_like(out)
        hip.rmsnorm_mod(_ld(x), same, 1e-5, **kw)                 # == the fp32 kernel on the values the halves stand for
        assert torch.equal(out, same)


@pytest.mark.parametrize("M,N,K", [(70000, 2560, 2560), (16300, 4096, 192), (9000, 2560, 6912)])
def test_gemm_persistent_h16_stream_epilogues(hip, ref, M, N, K):
    """The persistent GEMM kernel's two h16 forms (the NaDiT's residual stream): bias -> h16 (patch-in) and gate * (acc + bias) + h16
    residual -> h16 in place (attn-out / mlp-out), with and without the fragment-ordered weights; routed to the persistent kernel,
    <= 1e-3 like any wide store, bit-identical between the two main loops and from launch to launch."""
    packing = sub("packing")
    A = rnd(M, K)
    W = packing.pack_matrix(rnd(N, K, scale=1.0 / math.sqrt(K), seed=1), "cuda")
    Wf = hip.pack_gemm_frag(W)
    bias, gate = rnd(N, dtype=F32, seed=3), rnd(N, dtype=F32, seed=4)
    hid0 = (rnd(M, N, seed=6, dtype=F32) * 3.0 * H16_SCALE).to(H16)
    hip.record_kernel_class = True
    try:
        out = torch.full((M, N), float("nan"), device="cuda", dtype=H16)
        hip.gemm(A, W, out, N=N, K=K, bias=bias, out_f32=True, W_frag=Wf)
        assert hip.last_kernel_class == "gemm_persistent"
        assert rel_err(_ld(out), ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias)) < TOL_F32
        want = ref.gemm(A, W, torch.empty(M, N, device="cuda"), N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=hid0)
        outs = []
        for frag in (Wf, None, Wf):
            hid = hid0.clone()
            hip.gemm(A, W, hid, N=N, K=K, bias=bias, epilogue=EPI_RESID_GATE, gate=gate, resid=hid, out_f32=True, W_frag=frag)
            assert hip.last_kernel_class == "gemm_persistent"
            outs.append(hid)
        assert rel_err(_ld(outs[0]), want) < TOL_F32
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    finally:
        hip.record_kernel_class = False


def test_dit_engine_h16_stream_matches_its_emulation_and_the_fp32_stream(hip):
    """NaDiTEngine(hid_store="h16") on the HIP path against the same host code over the torch double of the C ABI in the same regime,
    and against the fp32-stream engine (the two differ by the h16 roundings of the stream only: ~3e-4 per store)."""
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    cfg = config.DIT_TINY
    sd = weights.synth_dit_state_dict(cfg, seed=5)
    g = torch.Generator().manual_seed(1)
    vid = torch.randn(3, 16, 24, 33, generator=g).to(BF16)
    txt = weights.synth_text_embedding()
    outs = {}
    for store in ("h16", "fp32", "bf16"):
        eng = dit.NaDiTEngine(cfg, sd, hip, hid_store=store)
        assert eng.hid_store == store
        outs[store] = eng.forward(vid.cuda(), txt.cuda(), 1000.0).float().cpu()
    emu = dit.NaDiTEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), hid_store="h16").forward(vid, txt, 1000.0).float()
    e_emu, e_32, e_16 = rel_err(outs["h16"], emu), rel_err(outs["h16"], outs["fp32"]), rel_err(outs["bf16"], outs["fp32"])
    print(f"DiT(tiny) h16 stream: vs its CPU emulation {e_emu:.3e}, vs the fp32 stream {e_32:.3e} (bf16 stream vs fp32: {e_16:.3e})")
    assert e_emu < 8e-3 and e_32 < e_16 and e_32 < 5e-3
