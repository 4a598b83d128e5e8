"""Group a rocprofv3 kernel trace by (kernel, grid, workgroup): separates the shapes one kernel template serves."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0, 0.0])
total = 0.0
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:64]
    key = (name, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    acc[key][0] += 1
    acc[key][1] += d
    total += d
print(f"# {len(rows)} dispatches, {total:.1f} ms of kernel time; kernel | grid x,y | wg | calls | total ms | avg us | share")
for key, (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f"{key[0]:66s} {key[1]:>10s} {key[2]:>6s} {key[3]:>5s} {n:6d} {ms:10.1f} {ms / n * 1e3:10.1f} {100 * ms / total:6.2f}%")
