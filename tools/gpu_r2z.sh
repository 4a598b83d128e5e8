#!/bin/bash
# Round 2: svr_gemm8.hip as shipped (opt-in, ablation hooks compiled out) -- parity
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 45 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "gemm8 or gemm_bias or gemm_epilogues" > gpurun_out/r2zz_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r2zz_pytest.log | tail -3
