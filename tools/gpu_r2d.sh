#!/bin/bash
# Round 2: conv fragment-prefetch schedule (conv_pf) -- parity, kbench A/B, bench cfg3 A/B on one box
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv3d" -x > gpurun_out/r2d_pytest_conv.log 2>&1
echo "pytest conv rc=$?"; tail -5 gpurun_out/r2d_pytest_conv.log
for pf in 0 1 2; do
  SVR_CONV_PF=$pf timeout 300 python tools/kbench.py --only conv --reps 5 > gpurun_out/r2d_kbench_pf$pf.jsonl 2> gpurun_out/r2d_kbench_pf$pf.err
  echo "kbench pf=$pf rc=$?"; python - <<PY
import json
print(' | '.join(f"{json.loads(l)['tflops']:.0f}" for l in open('gpurun_out/r2d_kbench_pf$pf.jsonl') if l.startswith('{')))
PY
done
for pf in 1 0; do
  SVR_CONV_PF=$pf timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2d_bench_pf$pf.json 2> gpurun_out/r2d_bench_pf$pf.err
  echo "bench pf=$pf rc=$?"; cat gpurun_out/r2d_bench_pf$pf.json; tail -2 gpurun_out/r2d_bench_pf$pf.err
done
