#!/bin/bash
# One gpurun call: full bench of a workload (+ optional full gpu test suite first)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WL=${1:-cfg3}; TESTS=${2:-0}; EXTRA=${3:-}
if [ "$TESTS" = "1" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
fi
timeout 1500 python bench.py --workload $WL --steps 2 --warmup 1 $EXTRA > gpurun_out/bench_$WL.json 2> gpurun_out/bench_$WL.err
echo "bench $WL rc=$?"; cat gpurun_out/bench_$WL.json; tail -5 gpurun_out/bench_$WL.err
