#!/usr/bin/env python
"""Static VALU issue-cycle mix of a kernel region, from the device assembly hipcc emits (no GPU needed).

    python tools/static_valu_mix.py conv_halo2_kernelILi16ELi3ELi0 [--region epilogue|prologue|all] [-D SOME_EXPERIMENT=1 ...] [--asm file.s]

Region ``epilogue`` = everything after the kernel's last MFMA, ``prologue`` = everything before its first (kernels without MFMAs: use
``all``).  Every VALU instruction is priced at its issue cost for a wave64 on a 16-lane SIMD: 4 cycles, 16 for the quarter-rate ones --
transcendentals (v_exp / v_rcp / v_rsq / v_sqrt / v_log / v_sin / v_cos) and 32-bit integer multiplies (v_mul_lo / v_mul_hi / v_mad_u64_u32
/ v_mad_i64_i32).  Counts are STATIC over all code paths of the region (every compiled epilogue body of a kernel is in it once), so they
compare builds of the same source -- e.g. the product against an experiment build with extra -D defines -- rather than predict a run time.
Used for DESIGN.md section 8 (profiles/r4_conv_epilogue_static_mix.txt)."""
import argparse
import collections
import importlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
QUARTER_INT = ("v_mul_lo_", "v_mul_hi_", "v_mad_u64_u32", "v_mad_i64_i32")
QUARTER_TR = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")


def device_asm(defines):
    hip_lib = importlib.import_module("comfyui-seedvr2_videoupscaler_amd.hip_lib")
    out = os.path.join(tempfile.mkdtemp(prefix="svr_asm_"), "svr_api.s")
    cmd = ["/opt/rocm/bin/hipcc"] + [f for f in hip_lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + [f"-D{d}" for d in defines] + \
          ["-S", "--cuda-device-only", os.path.join(hip_lib.CSRC, "svr_api.hip"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read()


def classify(op):
    if op.startswith(QUARTER_INT): return "int32 multiply (quarter rate)", 16
    if op.startswith(QUARTER_TR): return "transcendental (quarter rate)", 16
    if op.startswith("v_cvt"): return "conversions", 4
    if op.startswith("v_pk_"): return "packed fp32", 4
    if op.startswith("v_accvgpr"): return "accumulator moves", 4
    if re.search(r"_f(32|64|16)", op): return "scalar fp", 4
    return "integer / moves / selects", 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--region", default="epilogue", choices=["epilogue", "prologue", "all"])
    ap.add_argument("-D", dest="defines", action="append", default=[])
    ap.add_argument("--asm")
    args = ap.parse_args()
    s = open(args.asm).read() if args.asm else device_asm(args.defines)
    meta = s[s.rfind("amdhsa.kernels"):]
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n", s, re.M):
        name = m.group(1)
        if args.kernel not in name:
            continue
        lines = [l.split(";")[0].strip() for l in s[m.end():s.index(".Lfunc_end", m.end())].splitlines()]
        lines = [l for l in lines if l and not l.endswith(":")]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        if args.region == "epilogue" and mf:
            lines = lines[mf[-1] + 1:]
        elif args.region == "prologue" and mf:
            lines = lines[:mf[0]]
        cyc, cnt = collections.Counter(), collections.Counter()
        for l in lines:
            op = re.match(r"([a-z_0-9]+)", l)
            if op and op.group(1).startswith("v_") and not op.group(1).startswith("v_mfma"):
                c, w = classify(op.group(1))
                cyc[c] += w
                cnt[c] += 1
        blk = [b for b in re.split(r"\n\s*- \.agpr_count:", meta)[1:] if name in b]
        regs = ""
        if blk:
            g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk[0]).group(1)
            regs = f"  [{g('vgpr_count')} VGPRs, {g('vgpr_spill_count')} spilled, {g('private_segment_fixed_size')} B scratch]"
        total = sum(cyc.values())
        print(f"{name}\n  region {args.region}: {len(lines)} instructions, {total} VALU issue cycles{regs}" +
              (f"  (defines: {' '.join(args.defines)})" if args.defines else ""))
        for c, v in cyc.most_common():
            print(f"    {c:34s}{cnt[c]:7d} instr {v:8d} cycles {100.0 * v / max(total, 1):5.1f} %")


if __name__ == "__main__":
    main()
