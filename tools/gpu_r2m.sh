#!/bin/bash
# Round 2: sub-pixel conv kernel (svr_conv_sub.hip) -- parity, race screen, bench cfg3 A/B (conv_sub 1 vs 0)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -s -k "phase_scatter or subpixel or vae or pipeline" -x > gpurun_out/r2m_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "rel-err|PSNR" gpurun_out/r2m_pytest.log | tail -12; tail -4 gpurun_out/r2m_pytest.log
OPT_A=conv_sub=1 OPT_B=conv_sub=0 bash tools/gpu_bench_ab.sh
