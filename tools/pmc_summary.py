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
