#!/bin/bash
# Round 2: bench lines of the other workloads on the final build (7B cfg5, cfg4 on one GPU)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 500 python bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2r_bench_cfg5.json 2> gpurun_out/r2r_bench_cfg5.err
echo "bench cfg5 rc=$?"; cut -c1-250 gpurun_out/r2r_bench_cfg5.json
timeout 500 python bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2r_bench_cfg4.json 2> gpurun_out/r2r_bench_cfg4.err
echo "bench cfg4 rc=$?"; cut -c1-250 gpurun_out/r2r_bench_cfg4.json
