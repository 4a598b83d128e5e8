#!/bin/bash
# Round 2: first run of the GEMM kernel with register-streamed weights (svr_gemm8.hip): parity, then rate against gemm_kernel
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "gemm8" > gpurun_out/r2w_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r2w_pytest.log | tail -5
for o in 1 0; do SVR_OPTIONS=gemm_impl=$o timeout 60 python tools/kbench.py --only gemm --reps 3 2>/dev/null | tee -a gpurun_out/r2w_kbench.jsonl; done
