#!/bin/bash
# kbench under several env settings (measurement-only ablations of the pipelined GEMM).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ONLY=${1:-gemm}; shift
for v in "$@"; do
  echo "== SVR_PIPE_ABL=$v"
  SVR_PIPE_ABL=$v timeout 600 python tools/kbench.py --reps 5 --only $ONLY 2>/dev/null | tee gpurun_out/kbench_abl$v.jsonl
done
