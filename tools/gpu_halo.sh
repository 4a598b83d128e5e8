#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "conv" > gpurun_out/pytest_kernels.log 2>&1
echo "pytest conv rc=$?"; tail -5 gpurun_out/pytest_kernels.log
for v in "$@"; do echo "== ABL=$v"; SVR_PIPE_ABL=$v timeout 600 python tools/kbench.py --reps 5 --only conv 2>/dev/null | tee gpurun_out/kbench_abl$v.jsonl; done
