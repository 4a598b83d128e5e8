#!/bin/bash
# Round 2: LDS-staged GEMM epilogue -- parity (both paths), kbench A/B on DiT GEMMs and the VAE's short-K shapes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "gemm or pixel_shuffle or conv3d_implicit" -x > gpurun_out/r2f_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r2f_pytest.log
for epi in 1 2; do
  SVR_GEMM_EPI=$epi timeout 300 python tools/kbench.py --only gemm,shortk --reps 5 > gpurun_out/r2f_kbench_epi$epi.jsonl 2> gpurun_out/r2f_kbench_epi$epi.err
  echo "kbench epi=$epi rc=$?"; tail -2 gpurun_out/r2f_kbench_epi$epi.err
done
python - <<PY
import json
a=[json.loads(l) for l in open('gpurun_out/r2f_kbench_epi1.jsonl') if l.startswith('{')]
b=[json.loads(l) for l in open('gpurun_out/r2f_kbench_epi2.jsonl') if l.startswith('{')]
for x,y in zip(a,b):
    print(f"{x['kernel']:62s} direct {x['us']:9.1f} us {x.get('tflops',0):7.1f} TF | lds {y['us']:9.1f} us {y.get('tflops',0):7.1f} TF {y.get('gbps',0):7.0f} GB/s  x{x['us']/y['us']:.2f}")
PY
