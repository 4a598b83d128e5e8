#!/usr/bin/env python
"""List the s_waitcnt instructions hipcc's waitcnt pass put inside loops (i.e. NOT the ones our inline asm wrote): a hand-scheduled
K loop with counted waits is defeated by one compiler-inserted vmcnt(0) in its body (gemm_w4p_kernel carried one per K tile for a
while, profiles/r3_gemm_w4_ablations.txt section 7).   python tools/asm_compiler_waits.py [svr_api.s]   (hipcc -S --cuda-device-only)"""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/svr_api.s"
kern, in_asm, depth_hdr, out = None, False, None, {}
for ln, line in enumerate(open(path), 1):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern, depth_hdr = m.group(1), None
    if "s_endpgm" in line:
        kern = None
    if kern is None:
        continue
    if "#ASMSTART" in line:
        in_asm = True
    elif "#ASMEND" in line:
        in_asm = False
    elif "Loop Header" in line or "in Loop:" in line:
        depth_hdr = line.strip()
    elif re.match(r"^\.LBB\d+_\d+:\s*$", line):
        depth_hdr = None                                  # a label without a loop comment: outside loops
    elif "s_waitcnt" in line and not in_asm and depth_hdr:
        out.setdefault(kern, []).append((ln, line.strip(), depth_hdr[:60]))
for k, v in out.items():
    print(k[:90], len(v))
    for ln, w, h in v[:12]:
        print(f"   line {ln}: {w:34s} {h}")
