#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "7b or epilogues" -s 2>&1 | tail -8
timeout 1500 python bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err
echo "bench cfg5 rc=$?"; cut -c1-1200 gpurun_out/bench_cfg5.json; tail -3 gpurun_out/bench_cfg5.err
