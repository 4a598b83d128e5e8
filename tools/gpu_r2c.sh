#!/bin/bash
# Round 2: full GPU suite + smoke + bench cfg3 + bench cfg4 (1 GPU)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 -s > gpurun_out/r2c_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "rel-err|PSNR" gpurun_out/r2c_pytest_gpu.log | tail -20; tail -14 gpurun_out/r2c_pytest_gpu.log
timeout 400 python __graft_entry__.py smoke > gpurun_out/r2c_smoke.log 2>&1
echo "smoke rc=$?"; tail -4 gpurun_out/r2c_smoke.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/r2c_bench_cfg3.json 2> gpurun_out/r2c_bench_cfg3.err
echo "bench cfg3 rc=$?"; cat gpurun_out/r2c_bench_cfg3.json; tail -3 gpurun_out/r2c_bench_cfg3.err
timeout 900 python bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2c_bench_cfg4.json 2> gpurun_out/r2c_bench_cfg4.err
echo "bench cfg4 rc=$?"; cat gpurun_out/r2c_bench_cfg4.json; tail -3 gpurun_out/r2c_bench_cfg4.err
