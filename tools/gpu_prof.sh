#!/bin/bash
# rocprofv3 kernel trace + stats of one bench run; only the stats csv is pulled back
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WL=${1:-cfg3}; TAG=${2:-r1}
rm -rf gpurun_out/prof_${WL}_$TAG
timeout 1500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${WL}_$TAG -o prof --output-format csv -- \
    python bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_${WL}_$TAG.json 2> gpurun_out/prof_${WL}_$TAG.err
echo "prof rc=$?"; cut -c1-600 gpurun_out/prof_${WL}_$TAG.json
python tools/trace_by_shape.py $(find gpurun_out/prof_${WL}_$TAG -name "*kernel_trace.csv" | head -1) > gpurun_out/prof_${WL}_${TAG}_by_shape.txt 2>&1
find gpurun_out/prof_${WL}_$TAG -name "*kernel_trace.csv" -delete
python - <<PY
import csv,re
rows=list(csv.DictReader(open('gpurun_out/prof_${WL}_$TAG/prof_kernel_stats.csv')))
for r in rows[:28]:
    n=re.sub(r'\(.*','',r['Name']).replace('void ','')[:70]
    print(f"{n:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.1f} ms {float(r['AverageNs'])/1e3:10.1f} us {r['Percentage']}%")
PY
