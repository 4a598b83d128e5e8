#!/bin/bash
# One gpurun call: slicing debug, full-size bench (cfg3) and a rocprofv3 kernel trace of a bench run.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WL=${1:-cfg3}
PROF_WL=${2:-cfg2}
timeout 300 python tools/debug_slicing.py > gpurun_out/debug_slicing.log 2>&1
echo "debug rc=$?"; tail -70 gpurun_out/debug_slicing.log
timeout 1500 python bench.py --workload $WL --steps 1 --warmup 1 > gpurun_out/bench_$WL.json 2> gpurun_out/bench_$WL.err
echo "bench $WL rc=$?"; cat gpurun_out/bench_$WL.json; tail -5 gpurun_out/bench_$WL.err
rm -rf gpurun_out/prof_$PROF_WL
timeout 1200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$PROF_WL -o prof --output-format csv -- \
    python bench.py --workload $PROF_WL --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$PROF_WL.json 2> gpurun_out/prof_$PROF_WL.err
echo "prof rc=$?"; cat gpurun_out/prof_$PROF_WL.json; tail -5 gpurun_out/prof_$PROF_WL.err
find gpurun_out/prof_$PROF_WL -name "*kernel_trace.csv" -size +20M -delete   # keep the merge under the 64 MiB cap
find gpurun_out/prof_$PROF_WL -type f | head; 
f=$(find gpurun_out/prof_$PROF_WL -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
