#!/bin/bash
# Round 2: colour-fix wavelet blur without library convs -- pipeline parity on the GPU + cfg4 clip time
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -s -k "pipeline or sharded" 2>&1 | tail -5
timeout 400 python bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2u_cfg4.json 2> gpurun_out/r2u_cfg4.err
echo "cfg4 rc=$?"; cut -c1-260 gpurun_out/r2u_cfg4.json
