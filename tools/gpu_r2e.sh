#!/bin/bash
# Round 2: one-workgroup-per-CU probe of both conv schedules + kernel trace of the cfg3 bench grouped by shape
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
for pf in 0 1; do
  SVR_CONV_LDS=100000 SVR_CONV_PF=$pf timeout 300 python tools/kbench.py --only conv --reps 5 > gpurun_out/r2e_kbench_solo_pf$pf.jsonl 2> gpurun_out/r2e_kbench_solo_pf$pf.err
  echo "kbench solo pf=$pf rc=$?"; python - <<PY
import json
print(' | '.join(f"{json.loads(l)['tflops']:.0f}" for l in open('gpurun_out/r2e_kbench_solo_pf$pf.jsonl') if l.startswith('{')))
PY
done
bash tools/gpu_prof.sh cfg3 r2
head -60 gpurun_out/prof_cfg3_r2_by_shape.txt | cut -c1-150
