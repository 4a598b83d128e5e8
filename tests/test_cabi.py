"""-m "not gpu": the C-ABI library builds for gfx950, loads, and exports every symbol that
include/seedvr2_hip.h declares (no compute calls without a GPU); the product refuses to run without it."""
import ctypes
import os
import re
import shutil

import pytest
import torch

from conftest import sub, ROOT

HAVE_HIPCC = shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")


def _declared():
    src = open(os.path.join(ROOT, "include", "seedvr2_hip.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|int32_t|int64_t|const char\*)\s+(svr_\w+)\s*\(", src, flags=re.M)))


@pytest.mark.skipif(not HAVE_HIPCC, reason="hipcc not available")
def test_library_builds_loads_and_exports_header_symbols():
    hip_lib = sub("hip_lib")
    path = hip_lib.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    declared = _declared()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in seedvr2_hip.h but not exported"
    assert sorted(hip_lib.SYMBOLS) == declared, "ctypes table and header disagree"
    lib.svr_abi_version.restype = ctypes.c_int
    assert lib.svr_abi_version() == hip_lib.ABI_VERSION == 5
    assert hip_lib.lib().svr_abi_version() == 5
    # the binary carries the content hash of the sources it was compiled from; the loader refuses any other
    assert hip_lib.built_id() == hip_lib.source_id() and not hip_lib.needs_build()


def test_struct_layout_matches_header():
    hip_lib = sub("hip_lib")
    # svr_conv_geom: 18 int32 + 2 pointers; svr_pixel_shuffle: 7 int32
    assert ctypes.sizeof(hip_lib.ConvGeom) == 18 * 4 + 2 * 8
    assert ctypes.sizeof(hip_lib.PixelShuffle) == 7 * 4
    assert ctypes.sizeof(hip_lib.PhaseScatter) == 4 * 4 + 8 and hip_lib.GemmArgs.phase.offset % 8 == 0
    assert hip_lib.GemmArgs.conv.offset % 8 == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a GPU-less host")
def test_product_fails_loudly_without_gpu():
    ops, hip_lib = sub("ops"), sub("hip_lib")
    with pytest.raises(hip_lib.HipLibraryError):
        ops.HipOps("cpu")
    with pytest.raises(Exception):
        ops.HipOps("cuda:0")


def test_svr_options_environment_reaches_the_library():
    """SVR_OPTIONS="key=value,..." is applied through svr_set_option when the library is loaded (the measurement scripts'
    only channel: round 2 found getenv() calls inside static-initialiser lambdas of the .hip sources reading the WRONG
    variable, which silently turned three A/B runs into A/A runs); an unknown key or a malformed item is an error."""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import ctypes, importlib, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "m = importlib.import_module('comfyui-seedvr2_videoupscaler_amd.hip_lib')\n"
        "L = m.lib()\n"
        "print(ctypes.c_int.in_dll(L, '_ZN3svr11g_conv_rowsE').value, ctypes.c_int.in_dll(L, '_ZN3svr10g_gemm_epiE').value,"
        " ctypes.c_int.in_dll(L, '_ZN3svr14g_conv_lds_dbgE').value)\n")
    env = dict(os.environ, SVR_OPTIONS="conv_rows=8, gemm_epi=2,conv_lds=100000")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["8", "2", "100000"]
    env.pop("SVR_OPTIONS")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split()[1:] == ["0", "0"], (r.stdout, r.stderr)      # defaults
    for bad in ("no_such_key=1", "conv_rows"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(env, SVR_OPTIONS=bad), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "SVR_OPTIONS" in r.stderr


@pytest.mark.skipif(not HAVE_HIPCC, reason="hipcc not available")
def test_loader_refuses_a_binary_built_from_other_sources(tmp_path, monkeypatch):
    """svr_build_id() is the content hash of the sources the binary was compiled from: after any source edit (here: one more
    header in the source list) the library counts as stale -- build() would recompile, lib() refuses to load it."""
    hip_lib = sub("hip_lib")
    hip_lib.build()
    extra = tmp_path / "svr_extra.h"
    extra.write_text("// an edit\n")
    real = hip_lib.sources
    monkeypatch.setattr(hip_lib, "sources", lambda: real() + [str(extra)])
    assert hip_lib.needs_build()
    monkeypatch.setattr(hip_lib, "_lib", None)
    with pytest.raises(hip_lib.HipLibraryError, match="other sources"):
        hip_lib.lib()
