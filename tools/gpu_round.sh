#!/bin/bash
# One gpurun call: kernel + parity tests, smoke, small benches.  Everything is logged under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== build check" | tee gpurun_out/round.log
timeout 600 python __graft_entry__.py build >> gpurun_out/round.log 2>&1
echo "== pytest -m gpu" | tee -a gpurun_out/round.log
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/round.log
tail -40 gpurun_out/pytest_gpu.log
echo "== smoke" | tee -a gpurun_out/round.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/round.log
tail -6 gpurun_out/smoke.log
for wl in "$@"; do
  echo "== bench $wl" | tee -a gpurun_out/round.log
  timeout 900 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err
  echo "bench $wl rc=$?" | tee -a gpurun_out/round.log
  cat gpurun_out/bench_$wl.json; tail -5 gpurun_out/bench_$wl.err
done
