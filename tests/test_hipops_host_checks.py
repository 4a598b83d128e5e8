"""-m "not gpu": the host side of ``ops.HipOps`` (layout checks, struct filling, pointer arithmetic) driven by the real engines
on CPU tensors with a RECORDING FAKE of the C ABI in place of libseedvr2_hip.so.

HipOps itself can only be constructed on a GPU box, so a typo in one of its argument checks would first show up at the
round-end GPU run.  Here the same class runs with ``lib`` replaced by an object whose every ``svr_*`` function returns 0 and
records its arguments: outputs are never computed (the engines see uninitialised memory), but every call the NaDiT and VAE
engines make passes through HipOps' validation exactly as on the GPU, and the recorded calls are checked against the
library's real host-side validation rules.
"""
import ctypes

import pytest
import torch

from conftest import sub


class FakeLib:
    """Every svr_* entry point: returns 0 (svr_gemm_kernel_class: GEMM; svr_gemm_gn_blocks: 0 = no fused statistics;
    svr_groupnorm_workspace_bytes: 16) and records (name, args)."""

    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        if not name.startswith("svr_"):
            raise AttributeError(name)

        def fn(*args):
            self.calls.append((name, args))
            return {"svr_groupnorm_workspace_bytes": 16, "svr_gemm_kernel_class": 1}.get(name, 0)
        return fn


def fake_hipops(lib):
    ops_mod = sub("ops")
    ops = object.__new__(ops_mod.HipOps)
    ops.device = torch.device("cpu")
    ops.lib = lib
    ops.device_info = "fake"
    ops.zeros = torch.zeros(64, dtype=torch.uint8)
    ops.record_kernel_class, ops.last_kernel_class = False, None
    ops._stream = lambda: ctypes.c_void_p(0)
    return ops


@pytest.fixture(scope="module")
def recorded():
    config, weights = sub("config"), sub("weights")
    lib = FakeLib()
    ops = fake_hipops(lib)
    torch.manual_seed(0)
    # NaDiT (reduced width; MM + shared blocks, regular + shifted windows), then the VAE engine: untiled and tiled, both directions
    cfg = config.DIT_TINY
    eng = sub("dit").NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg), ops, overflow_guard=False)    # (outputs are uninitialised memory here)
    vid = torch.randn(3, 16, 24, 33).to(torch.bfloat16)
    eng.forward(vid, weights.synth_text_embedding(), 1000.0, x_t=vid[..., :16].contiguous())
    for store in ("fp32", "bf16"):                        # the other storage kinds of the residual stream (default: h16)
        sub("dit").NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg), ops, hid_store=store, overflow_guard=False).forward(
            vid, weights.synth_text_embedding(), 1000.0)
    n_dit = len(lib.calls)
    vcfg = config.VAE_V3
    veng = sub("vae").VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg), ops, overflow_guard=False)    # (outputs are uninitialised memory here)
    x = torch.rand(3, 5, 48, 64).to(torch.bfloat16)
    lat = veng.encode(x)
    veng.decode(torch.randn_like(lat.float()).to(lat.dtype))
    veng.encode(x, tiled=True, tile_size=(32, 32), tile_overlap=(8, 8))
    veng.decode(torch.randn_like(lat.float()).to(lat.dtype), tiled=True, tile_size=(32, 32), tile_overlap=(8, 8))
    # the call patterns the first four do not reach: several temporal slices with their carried halos, a decode that skips trimmed
    # frames, the other storage regimes of the residual trunk (fp32 / bf16: other operand kinds in every GroupNorm / epilogue call),
    # the reference's two-step upsampler and three-tap head, and the 7B family's block (GELU MLP, rope3d tables, no output norm)
    x9 = torch.rand(3, 9, 32, 48).to(torch.bfloat16)
    lat9 = veng.encode(x9, frames_per_slice=4)
    veng.decode(torch.randn_like(lat9.float()).to(lat9.dtype), latents_per_slice=1, keep_frames=6)
    veng.decode(torch.randn_like(lat9.float()).to(lat9.dtype), tiled=True, tile_size=(32, 32), tile_overlap=(8, 8), keep_frames=7)
    for kw in (dict(trunk_store="fp32", branch_store="bf16"), dict(trunk_store="bf16"), dict(trunk_store="fp32", branch_store="fp32"),
               dict(merge_upsamplers=False, merge_causal_head=False)):
        v2 = sub("vae").VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg), ops, overflow_guard=False, **kw)
        l2 = v2.encode(x)
        v2.decode(torch.randn_like(l2.float()).to(l2.dtype))
    cfg7 = config.DIT_7B_TINY
    eng7 = sub("dit").NaDiTEngine(cfg7, weights.synth_dit_state_dict(cfg7), ops, overflow_guard=False)
    eng7.forward(vid, weights.synth_text_embedding(), 1000.0, x_t=vid[..., :16].contiguous())
    return lib.calls, n_dit


class RoutedFakeLib(FakeLib):
    """FakeLib whose ROUTING answers are the real library's: svr_gemm_kernel_class / svr_gemm_gn_blocks are pure host functions of
    libseedvr2_hip.so (no GPU needed), so HipOps takes the branches it takes on the GPU -- fused GroupNorm statistics (partial
    buffers, svr_groupnorm_reduce, the shared buffer of a sub-pixel upsampler's launches) and the four phases as ONE quad launch."""

    def __init__(self, real):
        super().__init__()
        self.real = real

    def __getattr__(self, name):
        if name in ("svr_gemm_kernel_class", "svr_gemm_gn_blocks"):
            real_fn = getattr(self.real, name)

            def fn(*args):
                self.calls.append((name, args))
                return real_fn(*args)
            return fn
        return super().__getattr__(name)


@pytest.fixture(scope="module")
def recorded_routed():
    hip_lib = sub("hip_lib")
    try:
        real = hip_lib.lib()
    except hip_lib.HipLibraryError as e:
        pytest.skip(f"libseedvr2_hip.so not built: {e}")
    config, weights = sub("config"), sub("weights")
    lib = RoutedFakeLib(real)
    ops = fake_hipops(lib)
    torch.manual_seed(0)
    vcfg = config.VAE_V3
    veng = sub("vae").VideoVAEEngine(vcfg, weights.synth_vae_state_dict(vcfg), ops, overflow_guard=False)
    x = torch.rand(3, 9, 64, 96).to(torch.bfloat16)
    lat = veng.encode(x, frames_per_slice=4)
    veng.decode(torch.randn_like(lat.float()).to(lat.dtype), latents_per_slice=1, keep_frames=7)
    veng.decode(torch.randn_like(lat.float()).to(lat.dtype), tiled=True, tile_size=(32, 48), tile_overlap=(8, 8))
    cfg = config.DIT_TINY
    eng = sub("dit").NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg), ops, overflow_guard=False)
    vid = torch.randn(3, 16, 24, 33).to(torch.bfloat16)
    eng.forward(vid, weights.synth_text_embedding(), 1000.0, x_t=vid[..., :16].contiguous())
    return lib.calls


def test_engines_pass_hipops_validation_on_the_routes_the_library_picks(recorded_routed):
    """Same engines, the library's own routing: the launches that fuse GroupNorm statistics and the quad phase launches go through
    HipOps' bookkeeping (partial buffers sized from svr_gemm_gn_blocks, svr_groupnorm_reduce, gn_shared) without tripping a check."""
    calls = recorded_routed
    names = [c[0] for c in calls]
    assert "svr_groupnorm_reduce" in names                        # fused statistics were taken and reduced
    gemms = [c[1][0]._obj for c in calls if c[0] == "svr_gemm_bf16"]
    assert any(a.gn_partial for a in gemms) and any(a.phase.enabled and a.phase.quad for a in gemms)
    assert any(a.W_frag for a in gemms)
    for name, a in calls:
        if name == "svr_groupnorm_reduce":
            T, nblk, groups = a[2], a[3], a[4]
            assert T > 0 and nblk > 0 and groups == 32, a


def test_engines_pass_hipops_validation_and_reach_every_entry_point(recorded):
    calls, n_dit = recorded
    names = {c[0] for c in calls}
    assert {"svr_gemm_bf16", "svr_rmsnorm_mod", "svr_ada_combine", "svr_qknorm_rope", "svr_attn_varlen", "svr_rows_mean",
            "svr_patchify", "svr_unpatchify_euler"} <= {c[0] for c in calls[:n_dit]}
    assert {"svr_gemm_bf16", "svr_groupnorm_stats", "svr_groupnorm_apply", "svr_blend_accumulate", "svr_blend_finalize",
            "svr_affine_slice"} <= {c[0] for c in calls[n_dit:]}
    assert "svr_gemm_pack_frag" in names and "svr_conv_pack_frag_taps" in names          # fragment-ordered weight copies are requested


def test_recorded_calls_satisfy_the_librarys_own_host_rules(recorded):
    """The rules of csrc/svr_api.hip, applied to what the engines actually pass (scalars only: pointers are opaque here)."""
    calls, _ = recorded
    for name, a in calls:
        if name == "svr_rmsnorm_mod":
            rows, dim, x_f32 = a[2], a[3], a[8]
            assert rows > 0 and dim % 8 == 0 and 0 < dim <= 4096 and x_f32 in (0, 1, 2), a
        elif name == "svr_groupnorm_apply":
            T, HW, Cc, groups, x_f32 = a[5], a[6], a[7], a[8], a[11]
            assert T > 0 and HW > 0 and Cc % 8 == 0 and Cc <= 512 and groups > 0 and Cc % groups == 0 and x_f32 in (0, 1, 2), a
            assert a[2] is not None and a[3] is not None and a[4] is not None
        elif name == "svr_groupnorm_stats":
            T, HW, Cc, groups = a[3], a[4], a[5], a[6]
            assert Cc % 8 == 0 and Cc <= 512 and 0 < groups <= 32 and Cc % groups == 0 and (Cc // groups) % 4 == 0 and 256 % (Cc // 8) == 0, a
        elif name == "svr_qknorm_rope":
            assert a[1] > 0 and a[2] > 0 and a[7] > 0 and 0 < a[8] * 3 <= 64, a
        elif name == "svr_rows_mean":
            assert a[2] > 0 and 0 < a[3] <= 65535 and a[4] % 8 == 0, a
        elif name == "svr_unpatchify_euler":
            assert a[1] >= 4 * a[7] and a[5] % 2 == 0 and a[6] % 2 == 0, a
        elif name == "svr_blend_accumulate":
            T, h, w, Cc, H, W, y0, x0 = a[5:13]
            assert y0 >= 0 and x0 >= 0 and y0 + h <= H and x0 + w <= W, a
        elif name == "svr_blend_finalize":
            assert a[6] <= a[5], a                                     # c_take <= C
        elif name == "svr_affine_slice":
            assert 0 < a[4] <= a[3], a


def test_hipops_rejects_bad_side_operands():
    ops = fake_hipops(FakeLib())
    bf = torch.bfloat16
    x, out = torch.zeros(4, 64, dtype=bf), torch.zeros(4, 64, dtype=bf)
    with pytest.raises(ValueError, match="scale"):
        ops.rmsnorm_mod(x, out, 1e-6, scale=torch.zeros(64, dtype=bf))                   # fp32 expected
    with pytest.raises(ValueError, match="shift"):
        ops.rmsnorm_mod(x, out, 1e-6, shift=torch.zeros(32))                             # wrong length
    with pytest.raises(ValueError, match="rmsnorm_mod"):
        ops.rmsnorm_mod(x.to(torch.float16), out[:2], 1e-6)                              # (h16 is a stream format since round 5; shapes must agree)
    with pytest.raises(ValueError, match="activation storage"):
        ops.rmsnorm_mod(x.to(torch.float64), out, 1e-6)
    with pytest.raises(ValueError, match="q / k norm weights"):
        ops.qknorm_rope(torch.zeros(4, 3 * 2 * 128, dtype=bf), 2, torch.zeros(4, 3, dtype=torch.int16), 0, torch.zeros(8, 21), torch.zeros(8, 21),
                        None, torch.zeros(128), 1e-6)
    g = torch.zeros(2, 4, 4, 128, dtype=bf)
    with pytest.raises(ValueError, match="gamma"):
        ops.groupnorm_apply(g, torch.zeros_like(g), torch.zeros(2, 32, 2, dtype=torch.float64), torch.zeros(64), torch.zeros(128), 32, 1e-6, True)
    with pytest.raises(ValueError, match="stats"):
        ops.groupnorm_apply(g, torch.zeros_like(g), torch.zeros(2, 32, 2), torch.zeros(128), torch.zeros(128), 32, 1e-6, True)
    with pytest.raises(ValueError, match="attn_varlen"):
        i32 = torch.zeros(4, dtype=torch.int32)
        ops.attn_varlen(torch.zeros(4, 3 * 2 * 128, dtype=bf), torch.zeros(4, 128, dtype=bf), i32, i32, i32[:2], 4, 2, 128, 0.1)
    with pytest.raises(ValueError, match="rows_mean"):
        ops.rows_mean(torch.zeros(12, 64, dtype=bf), torch.zeros(3, 64, dtype=bf), 3, 5)
    with pytest.raises(ValueError, match="wy"):
        ops.blend_accumulate(torch.zeros(1, 4, 4, 3, dtype=bf), torch.zeros(1, 8, 8, 3), torch.zeros(8, 8), torch.zeros(5), torch.zeros(4), 0, 0)
