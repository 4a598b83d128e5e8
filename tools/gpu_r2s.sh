#!/bin/bash
# Round 2: cfg4 (8 batches of 17 frames on one GPU) with and without the two-term causal head, same box
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
i=0
for f in "" "--three-tap-head"; do
  i=$((i+1))
  timeout 400 python bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline $f > gpurun_out/r2s_cfg4_$i.json 2> gpurun_out/r2s_cfg4_$i.err
  echo -n "cfg4 [$f] rc=$? "; python - <<PY
import json
d=json.load(open('gpurun_out/r2s_cfg4_$i.json'))
print(round(d['ms_per_step'],1), round(d['value'],3), d['roofline']['per_kernel'])
PY
done
