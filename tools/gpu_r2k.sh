#!/bin/bash
# Round 2: frame-inner banded tile order of the conv kernel (memory-side-cache reuse of the temporal taps) -- kbench A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv3d_halo_kernel_race or fused_groupnorm" -x 2>&1 | tail -2
for band in ${BANDS:-0 4 8 16 2 0}; do
  SVR_OPTIONS=conv_band=$band timeout 300 python tools/kbench.py --only conv --reps 5 > gpurun_out/r2k_kbench_$band.jsonl 2> gpurun_out/r2k_kbench_$band.err
  echo -n "kbench band=$band rc=$? : "; python - <<PY
import json
print(' | '.join(f"{json.loads(l)['tflops']:.0f}" for l in open('gpurun_out/r2k_kbench_$band.jsonl') if l.startswith('{')))
PY
done
