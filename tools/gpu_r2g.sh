#!/bin/bash
# Round 2: 8-rows-per-wave conv (conv_rows 8) and one-workgroup-per-CU probe -- kbench A/B + bench cfg3 A/B on one box
# (options travel through SVR_OPTIONS -> svr_set_option at library load)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
for opt in "conv_rows=4" "conv_rows=8" "conv_rows=4,conv_lds=100000"; do
  tag=$(echo $opt | tr ',=' '__')
  SVR_OPTIONS=$opt timeout 300 python tools/kbench.py --only conv --reps 5 > gpurun_out/r2g_kbench_$tag.jsonl 2> gpurun_out/r2g_kbench_$tag.err
  echo "kbench $opt rc=$?"; python - <<PY
import json
print(' | '.join(f"{json.loads(l)['tflops']:.0f}" for l in open('gpurun_out/r2g_kbench_$tag.jsonl') if l.startswith('{')))
PY
done
for rows in 8 4; do
  SVR_OPTIONS=conv_rows=$rows timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2g_bench_rows$rows.json 2> gpurun_out/r2g_bench_rows$rows.err
  echo "bench rows=$rows rc=$?"; tail -2 gpurun_out/r2g_bench_rows$rows.err | cut -c1-200
  python - <<PY
import json
d=json.load(open('gpurun_out/r2g_bench_rows$rows.json'))
print({k:d[k] for k in ('value','ms_per_step','dit_ms_per_step','vae_encode_ms','vae_decode_ms')}, d['roofline']['per_kernel']['conv_halo'])
PY
done
