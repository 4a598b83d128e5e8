#!/bin/bash
# One gpurun call: full GPU test suite, smoke, kernel micro-bench, bench of a workload.  Logs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WL=${1:-cfg3}
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -14 gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
timeout 600 python tools/kbench.py --reps 5 > gpurun_out/kbench.jsonl 2> gpurun_out/kbench.err
echo "kbench rc=$?"; cat gpurun_out/kbench.jsonl
timeout 1500 python bench.py --workload $WL --steps 2 --warmup 1 > gpurun_out/bench_$WL.json 2> gpurun_out/bench_$WL.err
echo "bench $WL rc=$?"; cat gpurun_out/bench_$WL.json; tail -3 gpurun_out/bench_$WL.err
