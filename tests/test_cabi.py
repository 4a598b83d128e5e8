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
    assert lib.svr_abi_version() == 3
    assert hip_lib.lib().svr_abi_version() == 3


def test_struct_layout_matches_header():
    hip_lib = sub("hip_lib")
    # svr_conv_geom: 18 int32 + 2 pointers; svr_pixel_shuffle: 7 int32
    assert ctypes.sizeof(hip_lib.ConvGeom) == 18 * 4 + 2 * 8
    assert ctypes.sizeof(hip_lib.PixelShuffle) == 7 * 4
    assert hip_lib.GemmArgs.conv.offset % 8 == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a GPU-less host")
def test_product_fails_loudly_without_gpu():
    ops, hip_lib = sub("ops"), sub("hip_lib")
    with pytest.raises(hip_lib.HipLibraryError):
        ops.HipOps("cpu")
    with pytest.raises(Exception):
        ops.HipOps("cuda:0")
