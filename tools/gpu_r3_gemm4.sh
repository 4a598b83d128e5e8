#!/bin/bash
# Next round, first thing: gemm4_kernel (two workgroups per CU, svr_gemm8.hip) has never run -- parity, then rate against gemm8 / gemm_kernel
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "gemm8" > gpurun_out/r3_gemm4_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3_gemm4_pytest.log | tail -5
for o in 3 1 0; do echo "gemm_impl=$o"; SVR_OPTIONS=gemm_impl=$o timeout 60 python tools/kbench.py --only gemm --reps 3 2>/dev/null | tee -a gpurun_out/r3_gemm4_kbench.jsonl; done
for st in 1 2; do echo "gemm8 stagger=$st"; SVR_OPTIONS=gemm_impl=1,gemm_stagger=$st timeout 60 python tools/kbench.py --only gemm --reps 3 2>/dev/null | tee -a gpurun_out/r3_gemm4_kbench.jsonl; done
