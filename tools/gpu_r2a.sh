#!/bin/bash
# Round 2, first GPU call: new window-attention kernel (parity + micro-bench), fp32-store parity cases, new production-width goldens.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
# bail out early on a sick box (seen once: every process aborted at its first allocation, 10 GPU-minutes burnt)
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -x -k "attn or fp32_store or epilogues or swiglu" > gpurun_out/r2a_kernels.log 2>&1
echo "kernels rc=$?"; tail -15 gpurun_out/r2a_kernels.log
timeout 400 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -s -k "multiwindow or 17_frames or real_tile" > gpurun_out/r2a_parity.log 2>&1
echo "parity rc=$?"; grep -E "rel-err|passed|failed|Error|error" gpurun_out/r2a_parity.log | tail -20
timeout 300 python tools/kbench.py --reps 5 --only attn > gpurun_out/r2a_kbench_attn.jsonl 2> gpurun_out/r2a_kbench_attn.err
echo "kbench rc=$?"; cat gpurun_out/r2a_kbench_attn.jsonl; tail -3 gpurun_out/r2a_kbench_attn.err
