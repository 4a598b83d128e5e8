"""-m "not gpu": static resource checks of the hand-scheduled kernels (hipcc cross-compiles without a GPU).

The LDS-halo conv kernels issue loads as inline asm and count ``vmcnt`` by hand: a register spill (scratch traffic is
VMEM) or a register budget above 256 (one wave per SIMD instead of two) would silently break the schedule, so the
kernel metadata is pinned here."""
import os
import re
import shutil
import struct
import subprocess

import pytest

from conftest import ROOT, sub

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _hipcc():
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    return HIPCC if os.path.exists(HIPCC) else "hipcc"


def _compile_device_asm(tmp_path_factory, tag, extra):
    """Device-only assembly of csrc/svr_api.hip with the library's flags (+ ``extra``): (returncode, stderr, asm text)."""
    hip_lib = sub("hip_lib")
    out = tmp_path_factory.mktemp("asm") / f"svr_api_{tag}.s"
    cmd = [_hipcc()] + [f for f in hip_lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + extra + \
          ["-S", "--cuda-device-only", os.path.join(hip_lib.CSRC, "svr_api.hip"), "-o", str(out)]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return proc.returncode, proc.stderr, out.read_text() if proc.returncode == 0 and out.exists() else ""


def _metadata_of_built_library(tmp_path_factory):
    """The amdhsa metadata (YAML note) of the gfx950 code object inside the in-tree libseedvr2_hip.so -- the binary that ships --
    or None when the library is missing / was built from other sources (hip_lib.built_id() != source_id()) or the LLVM tools
    are not there.  The code object sits in a clang offload bundle: magic, bundle count, (offset, size, triple) entries."""
    hip_lib = sub("hip_lib")
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    try:
        if not os.path.exists(readelf) or hip_lib.built_id() != hip_lib.source_id():
            return None
        blob = open(hip_lib.LIB_PATH, "rb").read()
    except (OSError, AttributeError, hip_lib.HipLibraryError):
        return None
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    i = blob.find(magic)
    if i < 0:
        return None
    n, = struct.unpack_from("<Q", blob, i + len(magic))
    p = i + len(magic) + 8
    for _ in range(n):
        off, size, ts = struct.unpack_from("<QQQ", blob, p)
        triple = blob[p + 24:p + 24 + ts].decode()
        p += 24 + ts
        if "gfx950" in triple and size > 0:
            co = tmp_path_factory.mktemp("co") / "libseedvr2_hip.gfx950.co"
            co.write_bytes(blob[i + off:i + off + size])
            r = subprocess.run([readelf, "--notes", str(co)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            return r.stdout if r.returncode == 0 and "amdhsa.kernels" in r.stdout else None
    return None


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """Text holding the amdhsa.kernels metadata of the PRODUCT build: read from the in-tree library when it was built from the
    sources next to it (__graft_entry__.build() compiles it from HEAD; the loader refuses any other) -- the shipped binary itself,
    in milliseconds -- else the device pass is compiled here (~2 minutes)."""
    notes = _metadata_of_built_library(tmp_path_factory)
    if notes is not None:
        return notes
    rc, err, asm = _compile_device_asm(tmp_path_factory, "product", [])
    assert rc == 0, err[-2000:]
    return asm


def _kernels(asm):
    """name -> metadata dict from the amdhsa.kernels YAML block (compiler assembly or llvm-readelf --notes: same keys)."""
    meta = {}
    for blk in re.split(r"\n\s*- \.agpr_count:", asm)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
        meta[name.group(1)] = {k: get(k) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                   "private_segment_fixed_size", "max_flat_workgroup_size")}
    return meta


def test_hand_scheduled_kernels_do_not_spill_and_keep_their_occupancy(device_asm):
    meta = _kernels(device_asm)
    halo = {k: v for k, v in meta.items() if "conv_halo2_kernel" in k}
    assert len(halo) >= 4, sorted(meta)[:10]           # <16,0> LDS weights, <8,1> / <16,3> register-streamed weights, <8,2> thin input
    for name, m in halo.items():
        # (SGPR spills go to VGPR lanes with v_writelane: no memory traffic, allowed)
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (name, m)
        if "conv_halo2_kernelILi16ELi3E" in name:         # 8 rows per wave: 256 accumulators, ONE wave per SIMD by design
            assert 256 < m["vgpr_count"] <= 512 and m["max_flat_workgroup_size"] == 256, (name, m)
        else:
            assert m["vgpr_count"] <= 256, (name, m)      # two waves per SIMD (512 registers per lane)
    four_rows = [v for k, v in halo.items() if "conv_halo2_kernelILi8ELi1ELi0E" in k]
    assert four_rows and four_rows[0]["max_flat_workgroup_size"] == 256
    assert any("conv_halo2_kernelILi16ELi3ELi0E" in k for k in halo)
    # the four-wave GEMM: 256 accumulators pinned to AGPRs by its inline-asm MFMAs, fragments in < 256 VGPRs, one wave per SIMD
    w4 = [v for k, v in meta.items() if "gemm_w4q_kernel" in k]
    assert w4 and w4[0]["vgpr_spill_count"] == 0 and w4[0]["private_segment_fixed_size"] == 0 and 256 < w4[0]["vgpr_count"] <= 512
    # ... and its round-4 successor (weights straight into registers, activations by LDS-DMA)
    w4r = {k: v for k, v in meta.items() if "gemm_w4r_kernel" in k}
    assert len(w4r) == 1, sorted(w4r)
    for name, m in w4r.items():
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0 and 256 < m["vgpr_count"] <= 512, (name, m)
    # decoder conv_out (round 5): counted vmcnt(10) over a three-deep LDS-DMA ring, 96 accumulators, one four-wave workgroup per CU
    t4 = {k: v for k, v in meta.items() if "conv_thinout4_kernel" in k}
    assert len(t4) == 1 and not any("conv_thinout16" in k for k in meta), sorted(t4)
    for name, m in t4.items():
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0 and m["vgpr_count"] <= 512 and m["max_flat_workgroup_size"] == 256, (name, m)


def test_window_attention_builds_keep_their_occupancy(device_asm):
    """attn_win_kernel<NW, PRIO>: every build must stay at two waves per SIMD (<= 256 registers) and none may spill."""
    aw = {k: v for k, v in _kernels(device_asm).items() if "attn_win_kernel" in k}
    assert len(aw) == 4, sorted(aw)
    for name, m in aw.items():
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0 and m["vgpr_count"] <= 256, (name, m)


def test_no_kernel_uses_scratch(device_asm):
    """None of the library's kernels may spill: every one of them is bandwidth- or MFMA-bound by design."""
    bad = {k: v for k, v in _kernels(device_asm).items() if k.startswith("_ZN3svr") and v["private_segment_fixed_size"]}
    assert not bad, bad


def test_measurement_build_compiles(tmp_path_factory, request):
    """The -DSVR_ABLATIONS build (measurement-only kernel variants behind svr_set_option("pipe_abl"), tools/conv_timeline.py)
    must keep compiling: its variants instantiate the hand-written inline asm with different surrounding code.  Device pass only
    (the variants live in device code; the host pass and the link add a minute and prove nothing more)."""
    job = getattr(request.config, "_svr_measurement_build", None)      # started at collection time (tests/conftest.py), collected here
    if job is not None:
        proc, out = job
        _, err = proc.communicate()
        rc, asm = proc.returncode, (open(out).read() if proc.returncode == 0 and os.path.exists(out) else "")
    else:
        rc, err, asm = _compile_device_asm(tmp_path_factory, "ablations", ["-DSVR_ABLATIONS"])
    assert rc == 0, err[-3000:]
    assert "conv_halo2_kernelILi16ELi3ELi256E" in asm and "gemm_w4p_kernelILb1ELi8E" in asm      # timeline / ablation variants
    assert "gemm_w4r_kernelILi4EE" in asm and "gemm_w4r_kernelILi64EE" in asm                     # K-loop ablations of gemm_w4r_kernel
    assert "conv_thinout4_kernelILi1EE" in asm and "conv_thinout4_kernelILi8EE" in asm            # staging / store ablations of conv_out
