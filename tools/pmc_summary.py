"""Summarise rocprofv3 --pmc counter CSVs: per kernel name, mean counter value per dispatch.
usage: pmc_summary.py <dir-prefix>   (reads <prefix>1, <prefix>2, ... /**/*counter_collection.csv)"""
import csv
import glob
import re
import sys
from collections import defaultdict

prefix = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, n]
for path in sorted(glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True)):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or "?"
            if "svr::" not in name:
                continue
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            a = agg[name][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k in sorted(agg):
    print(k)
    for c, (s, n) in sorted(agg[k].items()):
        print(f"    {c:36s} mean/dispatch {s / n:18.1f}   dispatches {n}")

# kernel durations of the same passes (kernel_trace.csv): mean ns per dispatch, and the implied shader clock when
# GRBM_GUI_ACTIVE (cycles the GPU was busy, counted once per dispatch) was collected
dur = defaultdict(lambda: [0.0, 0])
for path in sorted(glob.glob(prefix + "*/**/*kernel_trace.csv", recursive=True)):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or "?"
            if "svr::" not in name:
                continue
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            d = dur[name]
            d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            d[1] += 1
print("# durations under the counter passes (ns, mean per dispatch) and implied clock")
for k in sorted(dur):
    ns = dur[k][0] / max(dur[k][1], 1)
    line = f"{k:60s} {ns:14.0f} ns"
    if "GRBM_GUI_ACTIVE" in agg[k]:
        s, n = agg[k]["GRBM_GUI_ACTIVE"]
        line += f"   GRBM_GUI_ACTIVE/ns = {s / n / ns:6.3f} GHz (x XCD count if summed over XCDs)"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in agg[k] and "SQ_BUSY_CU_CYCLES" in agg[k]:
        line += f"   mfma_busy/cu_busy = {agg[k]['SQ_VALU_MFMA_BUSY_CYCLES'][0] / agg[k]['SQ_BUSY_CU_CYCLES'][0]:5.3f}"
    print(line)
