"""rocprofv3 --pmc passes of a bench run -> per-kernel HBM traffic JSON (what bench.py's roofline.traffic reads).
usage: pmc_traffic_json.py <dir-prefix> <workload> > profiles/rN_<workload>_pmc_traffic.json
Corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count KiB; on gfx950
FETCH_SIZE under-reports wide (16 B / lane) streaming reads by 2x -> bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

prefix, workload = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "cfg3")
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sorted(glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True)):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or "?"
            if "svr::" not in name:
                continue
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            a = agg[name][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
# per-kernel mean duration in the pass that collected GRBM_GUI_ACTIVE (the kernel trace of the same run): effective shader clock
dur = defaultdict(lambda: [0.0, 0])
for path in sorted(glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True)):
    with open(path, newline="") as f:
        if "GRBM_GUI_ACTIVE" not in f.read():
            continue
    for tr in glob.glob(path.rsplit("/", 1)[0] + "/*kernel_trace.csv"):
        with open(tr, newline="") as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name") or "?"
                if "svr::" not in name:
                    continue
                name = re.sub(r"\(.*", "", name).replace("void ", "")
                try:
                    d = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                except (KeyError, ValueError):
                    continue
                dur[name][0] += d
                dur[name][1] += 1
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC / SQ passes (separate, --kernel-trace only) -- python bench.py "
                 f"--workload {workload} --steps 1 --warmup 0 --no-cpu-baseline; tools/gpu_pmc_bench.sh",
       "corrections": "bytes = FETCH_SIZE*1024*2 (gfx950 wide-read under-count) + WRITE_SIZE*1024",
       "workload": workload, "kernels": {}}
mean = lambda k, c: (agg[k][c][0] / agg[k][c][1]) if agg[k][c][1] else None
best = (0.0, None)
for k in sorted(agg):
    f, w = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    if f is None or w is None:
        continue
    e = {"dispatches": agg[k]["FETCH_SIZE"][1], "fetch_bytes_per_launch": f * 2048.0, "write_bytes_per_launch": w * 1024.0,
         "hbm_bytes_per_launch": f * 2048.0 + w * 1024.0}
    h, m = mean(k, "TCC_HIT_sum"), mean(k, "TCC_MISS_sum")
    if h is not None and m is not None and h + m > 0:
        e["l2_hit_rate"] = h / (h + m)
    mb, bc = mean(k, "SQ_VALU_MFMA_BUSY_CYCLES"), mean(k, "SQ_BUSY_CU_CYCLES")
    if mb is not None and bc:
        e["mfma_busy_frac"] = mb / (4.0 * bc)            # 4 SIMDs per CU
    ga = mean(k, "GRBM_GUI_ACTIVE")
    if ga is not None:
        e["gui_active_cycles_per_launch_all_xcd"] = ga
        if dur[k][1]:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; ns of the same run's kernel trace -> GHz under this kernel's load
            e["avg_duration_ns_in_pmc_pass"] = dur[k][0] / dur[k][1]
            e["shader_clock_ghz"] = ga / 8.0 / (dur[k][0] / dur[k][1])
    out["kernels"][k] = e
    tot = e["hbm_bytes_per_launch"] * e["dispatches"]
    if "conv_halo2" in k and tot > best[0]:
        best = (tot, k)
if best[1]:
    out["dominant"] = best[1]
json.dump(out, sys.stdout, indent=1)
