#!/bin/bash
# one cfg3 bench step (1 warm-up) + optional extra args; prints the line and the phase split
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-quick}; shift || true
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"; tail -2 gpurun_out/bench_$TAG.err | cut -c1-300
python - <<PY
import json
d=json.load(open('gpurun_out/bench_$TAG.json'))
print({k:d[k] for k in ('value','ms_per_step','dit_ms_per_step','vae_encode_ms','vae_decode_ms')})
print(d['roofline']['per_kernel'], d['roofline']['achieved'])
PY
