#!/bin/bash
# Round 2 final: full GPU suite, smoke, bench lines (cfg3 with cpu_baseline / cfg5 / cfg4), PMC traffic passes of the cfg3 step
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=4 -s > gpurun_out/r2z_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "rel-err|PSNR" gpurun_out/r2z_pytest_gpu.log | tail -22; tail -8 gpurun_out/r2z_pytest_gpu.log
timeout 400 python __graft_entry__.py smoke > gpurun_out/r2z_smoke.log 2>&1
echo "smoke rc=$?"; tail -4 gpurun_out/r2z_smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r2z_bench_cfg3.json 2> gpurun_out/r2z_bench_cfg3.err
echo "bench cfg3 rc=$?"; cut -c1-300 gpurun_out/r2z_bench_cfg3.json
timeout 900 python bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2z_bench_cfg5.json 2> gpurun_out/r2z_bench_cfg5.err
echo "bench cfg5 rc=$?"; cut -c1-250 gpurun_out/r2z_bench_cfg5.json
timeout 900 python bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2z_bench_cfg4.json 2> gpurun_out/r2z_bench_cfg4.err
echo "bench cfg4 rc=$?"; cut -c1-250 gpurun_out/r2z_bench_cfg4.json
bash tools/gpu_pmc_bench.sh cfg3 r2z > gpurun_out/r2z_pmcb.log 2>&1; grep -A13 "conv_halo2_kernel<16, 3, 0>" gpurun_out/pmcb_r2z_summary.txt | head -14
find gpurun_out -name "*counter_collection.csv" -delete; find gpurun_out -name "*kernel_trace.csv" -delete
