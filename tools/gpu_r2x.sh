#!/bin/bash
# Round 2: what bounds svr_gemm8.hip -- run-time ablations (results invalid): 1 no weight loads | 2 no LDS-DMA | 4 no fragment reads | 8 no stores
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for o in 0 1 2 4 8 7; do echo "abl=$o"; SVR_OPTIONS=gemm_impl=1,pipe_abl=$o timeout 40 python tools/kbench.py --only gemm --reps 2 2>/dev/null | tee -a gpurun_out/r2x_kbench.jsonl; done
