#!/bin/bash
# Round 2: pipeline parity on the GPU after the colour-fix change (log kept)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -s -k "pipeline or sharded" > gpurun_out/r2v_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "PSNR|passed|failed|Error" gpurun_out/r2v_pytest.log | tail -6
