#!/bin/bash
# One gpurun call: kernel tests + kbench A/B (legacy one-barrier GEMM vs pipelined GEMM).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ONLY=${1:-conv,gemm}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k conv > gpurun_out/pytest_kernels.log 2>&1
echo "pytest kernels rc=$?"; tail -15 gpurun_out/pytest_kernels.log
SVR_CONV_IMPL=2 timeout 600 python tools/kbench.py --reps 5 --only $ONLY > gpurun_out/kbench_legacy.jsonl 2> gpurun_out/kbench_legacy.err
echo "legacy rc=$?"; cat gpurun_out/kbench_legacy.jsonl
timeout 600 python tools/kbench.py --reps 5 --only $ONLY > gpurun_out/kbench_new.jsonl 2> gpurun_out/kbench_new.err
echo "new rc=$?"; cat gpurun_out/kbench_new.jsonl; tail -3 gpurun_out/kbench_new.err
