#!/bin/bash
# Round 2: causal-head layer test + PMC traffic passes of the cfg3 step on the final kernels / launch list
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -s -k "causal_head or subpixel_upsampler" 2>&1 | tail -6
bash tools/gpu_pmc_bench.sh cfg3 r2q > gpurun_out/r2q_pmcb.log 2>&1; tail -5 gpurun_out/r2q_pmcb.log | cut -c1-200
grep -A13 "conv_halo2_kernel<16, 3, 0>" gpurun_out/pmcb_r2q_summary.txt | head -16
