#!/bin/bash
# measurement-only ablations of the 8-row conv kernel (libseedvr2_hip_abl.so): what does each operand path cost in place?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SVR_BUILD_ABLATIONS=1
for abl in ${ABLS:-0 16 1 64 17 81 0}; do
  SVR_OPTIONS="pipe_abl=$abl" timeout 300 python tools/kbench.py --only conv --reps 5 > gpurun_out/abl8_$abl.jsonl 2> gpurun_out/abl8_$abl.err
  echo -n "abl=$abl rc=$? : "; python - <<PY
import json
print(' | '.join(f"{json.loads(l)['tflops']:.0f}" for l in open('gpurun_out/abl8_$abl.jsonl') if l.startswith('{')))
PY
done 2>&1 | tee gpurun_out/abl8_summary.txt
