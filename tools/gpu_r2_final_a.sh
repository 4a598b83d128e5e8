#!/bin/bash
# Round 2 final measurements, part A: kernel bench, kernel trace of the cfg3 bench, bench lines (cfg3 / cfg5 / cfg4)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 600 python tools/kbench.py --reps 5 --only conv,gemm,shortk,attn,side --attn-default-only > gpurun_out/r2_kbench.jsonl 2> gpurun_out/r2_kbench.err
echo "kbench rc=$?"; tail -2 gpurun_out/r2_kbench.err | cut -c1-200; wc -l gpurun_out/r2_kbench.jsonl
bash tools/gpu_prof.sh cfg3 r2 > gpurun_out/r2_prof.log 2>&1; tail -3 gpurun_out/r2_prof.log | cut -c1-200
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r2_bench_cfg3_final.json 2> gpurun_out/r2_bench_cfg3_final.err
echo "bench cfg3 rc=$?"; cut -c1-330 gpurun_out/r2_bench_cfg3_final.json; tail -2 gpurun_out/r2_bench_cfg3_final.err | cut -c1-200
timeout 900 python bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_cfg5.json 2> gpurun_out/r2_bench_cfg5.err
echo "bench cfg5 rc=$?"; cut -c1-330 gpurun_out/r2_bench_cfg5.json
timeout 900 python bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_bench_cfg4.json 2> gpurun_out/r2_bench_cfg4.err
echo "bench cfg4 rc=$?"; cut -c1-330 gpurun_out/r2_bench_cfg4.json
