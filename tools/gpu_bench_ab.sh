#!/bin/bash
# cfg3 bench A/B of two SVR_OPTIONS settings on one box, alternating: OPT_A OPT_B OPT_A OPT_B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
i=0
for opt in "$OPT_A" "$OPT_B" "$OPT_A" "$OPT_B"; do
  i=$((i+1))
  SVR_OPTIONS=$opt timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err
  echo -n "[$opt] rc=$? "; python - <<PY
import json
d=json.load(open('gpurun_out/ab_$i.json'))
print({k:round(d[k],1) for k in ('ms_per_step','dit_ms_per_step','vae_encode_ms','vae_decode_ms')}, d['roofline']['per_kernel']['conv_halo'])
PY
done
