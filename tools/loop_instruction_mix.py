#!/usr/bin/env python
"""Static instruction mix of a kernel's hot loop, from the device assembly hipcc emits (no GPU needed).

    python tools/loop_instruction_mix.py attn_win_kernel            # every instantiation whose mangled name contains the text
    python tools/loop_instruction_mix.py conv_sub_kernel --ops      # + the opcode histogram of the loop

For each matching kernel: the loop (backward branch) that holds the most MFMAs, its instruction counts by issue class, and the
issue slots the VALU port needs per MFMA (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos are quarter rate: 4 slots).  A
v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32 cycles (16x16x32: 16), an ordinary VALU issue its port for 4: a loop whose
waves need more VALU cycles than MFMA cycles per SIMD is VALU-bound whatever its memory pipeline does.  Counts are STATIC: rarely
taken branches inside the loop (ragged-tile masks, deferred rescales) are counted as if always executed -- read the listing
(--dump) before quoting a number.  Used for DESIGN.md 3.3 (profiles/r4_attn_instruction_mix.txt)."""
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
QUARTER = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def device_asm() -> str:
    hip_lib = importlib.import_module("comfyui-seedvr2_videoupscaler_amd.hip_lib")
    out = os.path.join(tempfile.mkdtemp(prefix="svr_asm_"), "svr_api.s")
    cmd = ["/opt/rocm/bin/hipcc"] + [f for f in hip_lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + \
          ["-S", "--cuda-device-only", os.path.join(hip_lib.CSRC, "svr_api.hip"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read()


def classify(op: str) -> str:
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(QUARTER): return "valu_quarter_rate"
    if op.startswith("v_accvgpr"): return "accvgpr_move"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def hot_loop(body: str):
    lines = [l.split(";")[0].rstrip() for l in body.splitlines()]
    lines = [l for l in lines if l.strip()]
    labels = {l.strip()[:-1]: i for i, l in enumerate(lines) if l.strip().endswith(":")}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"\s*s_c?branch\w*\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            n = sum(1 for x in lines[a:i + 1] if "v_mfma" in x)
            if best is None or n > best[0]:
                best = (n, a, i)
    if best is None or best[0] == 0:
        return None
    return lines[best[1]:best[2] + 1]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("kernel", help="text the mangled kernel name must contain")
    ap.add_argument("--ops", action="store_true", help="opcode histogram of the loop")
    ap.add_argument("--dump", action="store_true", help="print the loop's assembly")
    args = ap.parse_args()
    asm = device_asm()
    for m in re.finditer(r"^(_Z\w*" + re.escape(args.kernel) + r"\w*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        loop = hot_loop(m.group(2))
        if loop is None:
            continue
        ops = [l.split()[0] for l in loop if not l.strip().startswith(".") and not l.strip().endswith(":")]
        c = collections.Counter(classify(o) for o in ops)
        mf = max(c["mfma"], 1)
        slots = c["valu"] + c["accvgpr_move"] + 4 * c["valu_quarter_rate"]
        mfma_cycles = sum(32 if "32x32" in o else 16 for o in ops if o.startswith("v_mfma"))
        print(f"{m.group(1)}\n  loop: {len(ops)} instructions  {dict(sorted(c.items()))}")
        print(f"  VALU-port issue slots {slots} (x4 = {4 * slots} cycles) against {mfma_cycles} matrix-pipe cycles per wave and iteration; "
              f"{slots / mf:.1f} slots per MFMA")
        if args.ops:
            hist = collections.Counter(o for o in ops if classify(o) in ("valu", "valu_quarter_rate", "accvgpr_move"))
            print("  " + ", ".join(f"{k} {v}" for k, v in hist.most_common(24)))
        if args.dump:
            print("\n".join(loop))


if __name__ == "__main__":
    main()
