#!/bin/bash
# Round 2: full GPU suite + smoke + bench cfg3 on the current defaults (8-row conv, LDS-staged GEMM epilogue)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=6 -s > gpurun_out/r2h_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "rel-err|PSNR" gpurun_out/r2h_pytest_gpu.log | tail -24; tail -12 gpurun_out/r2h_pytest_gpu.log
timeout 400 python __graft_entry__.py smoke > gpurun_out/r2h_smoke.log 2>&1
echo "smoke rc=$?"; tail -4 gpurun_out/r2h_smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r2h_bench_cfg3.json 2> gpurun_out/r2h_bench_cfg3.err
echo "bench cfg3 rc=$?"; cut -c1-400 gpurun_out/r2h_bench_cfg3.json; tail -3 gpurun_out/r2h_bench_cfg3.err | cut -c1-300
