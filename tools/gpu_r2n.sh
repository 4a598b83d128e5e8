#!/bin/bash
# Round 2: thin-output conv kernel -- parity + kbench A/B + bench A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "thin_output or conv3d_implicit or fp32_store" -x 2>&1 | tail -4
for o in 1 0; do echo -n "kbench conv_out conv_thinout=$o: "; SVR_OPTIONS=conv_thinout=$o timeout 300 python tools/kbench.py --only conv --match "conv_out" --reps 5 2>/dev/null; done
OPT_A=conv_thinout=1 OPT_B=conv_thinout=0 bash tools/gpu_bench_ab.sh
