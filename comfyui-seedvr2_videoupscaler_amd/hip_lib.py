"""ctypes binding of libseedvr2_hip.so (the C ABI declared in include/seedvr2_hip.h).

The library is the product: if it is missing or fails to load, every op raises -- there is no
CPU / eager-PyTorch fallback on purpose (a silent fallback would void every parity claim).
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# (SVR_BUILD_ABLATIONS=1 selects the measurement build, kept in its own file so that it never replaces the product library)
LIB_PATH = os.path.join(CSRC, "libseedvr2_hip_abl.so" if os.environ.get("SVR_BUILD_ABLATIONS") else "libseedvr2_hip.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include", "seedvr2_hip.h")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
               "-mllvm", "-pragma-unroll-threshold=1000000"]

EPI_BIAS, EPI_BIAS_SILU, EPI_RESID_GATE, EPI_SWIGLU, EPI_BIAS_GELU = 0, 1, 2, 3, 4
ABI_VERSION = 9
# svr_gemm_kernel_class() codes (include/seedvr2_hip.h)
KERNEL_CLASSES = {0: "none", 1: "gemm", 2: "gemm_persistent", 3: "conv_halo", 4: "conv_subpixel", 5: "conv_thin_in",
                  6: "conv_thinout", 7: "conv_generic"}


class ConvGeom(C.Structure):
    _fields_ = [("enabled", C.c_int32),
                ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("To", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("kt", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("st", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
                ("pt", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32),
                ("halo_frames", C.c_int32),
                ("halo", C.c_void_p), ("zeros", C.c_void_p)]


class PixelShuffle(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("F", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("rz", C.c_int32), ("C", C.c_int32), ("drop_first", C.c_int32)]


class PhaseScatter(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("py", C.c_int32), ("px", C.c_int32), ("t_stride", C.c_int32),
                ("bias_border", C.c_void_p), ("quad", C.c_int32), ("reserved_", C.c_int32),
                ("W_frag4", C.c_void_p * 4), ("bias4", C.c_void_p * 4), ("bias_border4", C.c_void_p * 4)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64),
                ("W", C.c_void_p),
                ("C", C.c_void_p), ("ldc", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("bias", C.c_void_p), ("gate", C.c_void_p),
                ("resid", C.c_void_p), ("ldr", C.c_int64),
                ("epilogue", C.c_int32), ("out_f32", C.c_int32),
                ("conv", ConvGeom), ("ps", PixelShuffle),
                ("gn_partial", C.c_void_p), ("gn_groups", C.c_int32),
                ("W_frag", C.c_void_p),
                ("phase", PhaseScatter),
                ("resid_f32", C.c_int32)]


# name -> (restype, argtypes); must list every symbol declared in include/seedvr2_hip.h
_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    "svr_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "svr_gemm_pack_frag": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svr_conv_pack_frag": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "svr_conv_pack_frag_taps": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "svr_gemm_gn_blocks": (C.c_int32, [C.POINTER(GemmArgs)]),
    "svr_gemm_kernel_class": (C.c_int32, [C.POINTER(GemmArgs)]),
    "svr_gemm_kernel_name": (C.c_char_p, [_i32]),
    "svr_groupnorm_reduce": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "svr_rmsnorm_mod": (C.c_int, [_vp, _vp, _i64, _i32, _f, _vp, _vp, _vp, _i32, _vp]),
    "svr_ada_combine": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "svr_qknorm_rope": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _f, _vp]),
    "svr_attn_varlen": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f, _vp]),
    "svr_softmax_rows": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _i64, _f, _vp]),
    "svr_rows_mean": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "svr_patchify": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "svr_unpatchify_euler": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "svr_groupnorm_workspace_bytes": (C.c_int64, [_i32, _i64, _i32]),
    "svr_groupnorm_stats": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _vp]),
    "svr_groupnorm_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f, _i32, _i32, _vp]),
    "svr_im2col_causal": (C.c_int, [_vp, _vp, C.POINTER(ConvGeom), _i32, _vp]),
    "svr_blend_accumulate": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "svr_blend_finalize": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _f, _f, _vp]),
    "svr_affine_slice": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _f, _f, _vp]),
    "svr_set_option": (C.c_int, [C.c_char_p, _i32]),
    "svr_mfma_calibrate_workspace_bytes": (C.c_int64, []),
    "svr_mfma_calibrate": (C.c_int, [_vp, _i32, C.POINTER(C.c_double), _vp]),
    "svr_last_error": (C.c_char_p, []),
    "svr_abi_version": (C.c_int, []),
    "svr_build_id": (C.c_char_p, []),
    "svr_device_info": (C.c_int, [C.c_char_p, _i32]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def sources():
    """Every file the library is compiled from: csrc/*.hip|.h, the measurement-only pieces under csrc/measure/ (compiled only with
    -DSVR_ABLATIONS; hashed always, so both kinds of build name the tree they came from) and the public header."""
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    measure = os.path.join(CSRC, "measure")
    if os.path.isdir(measure):
        files += [os.path.join(measure, f) for f in sorted(os.listdir(measure)) if f.endswith((".inc", ".hip", ".h"))]
    return files + [INCLUDE]


def _extra_defines():
    return ["-DSVR_ABLATIONS"] if os.environ.get("SVR_BUILD_ABLATIONS") else []   # measurement-only kernel variants


def source_id() -> str:
    """hex SHA-256 over the library's sources (names and contents, sorted by name) and its compile configuration (flags and
    defines -- a -DSVR_ABLATIONS measurement build must not pass for the product build): what svr_build_id() of a current
    build returns."""
    h = hashlib.sha256()
    h.update("\0".join(HIPCC_FLAGS + _extra_defines()).encode() + b"\0")
    for path in sources():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def built_id(path: str = None):
    """The build id of the binary at ``path``, read from the FILE (the marker svr_build_id() returns a pointer into): dlopen
    would hand back an already-loaded older mapping of the same path.  None: no such file / no marker."""
    import re
    try:
        with open(path or LIB_PATH, "rb") as f:
            m = re.search(rb"SVR_BUILD_ID=([0-9a-f]{64}|unknown)", f.read())
        return m.group(1).decode() if m else None
    except OSError:
        return None


def needs_build() -> bool:
    """True unless the binary was compiled from exactly the sources next to it (content hash embedded at build time -- file times
    say nothing after a `git checkout`, and a stale binary newer than the sources would otherwise load silently)."""
    return not os.path.exists(LIB_PATH) or built_id() != source_id()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into libseedvr2_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    extra = _extra_defines()
    cmd = [hipcc] + HIPCC_FLAGS + extra + [f'-DSVR_BUILD_ID="{source_id()}"', os.path.join(CSRC, "svr_api.hip"), "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise HipLibraryError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle.  Raises HipLibraryError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the SeedVR2 HIP kernels are not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no CPU fallback.")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name} (stale build?)") from e
        fn.restype, fn.argtypes = res, args
    if handle.svr_abi_version() != ABI_VERSION:
        raise HipLibraryError("libseedvr2_hip.so ABI version mismatch; rebuild")
    if handle.svr_build_id().decode() != source_id():
        raise HipLibraryError(f"{LIB_PATH} was built from other sources than the ones next to it (build id "
                              f"{handle.svr_build_id().decode()[:12]}, sources {source_id()[:12]}): rebuild with "
                              "`python -c 'import __graft_entry__ as g; g.build()'`")
    # measurement knobs from the environment, e.g. SVR_OPTIONS="conv_rows=8,gemm_epi=1" (svr_set_option keys; an unknown key
    # or a malformed item is an error, not a silently ignored setting)
    for item in filter(None, (t.strip() for t in os.environ.get("SVR_OPTIONS", "").split(","))):
        key, sep, val = item.partition("=")
        if not sep or handle.svr_set_option(key.strip().encode(), int(val)) != 0:
            raise HipLibraryError(f"SVR_OPTIONS: bad item {item!r}")
        OPTIONS[key.strip()] = int(val)
    _lib = handle
    return _lib


OPTIONS = {}      # knobs set so far through SVR_OPTIONS / HipOps.set_option (the library has no getter)


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().svr_last_error()
        raise HipLibraryError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
