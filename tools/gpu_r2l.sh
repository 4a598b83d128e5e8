#!/bin/bash
# Round 2: sub-pixel form of the VAE's spatial upsampler -- kernel + engine parity, VAE goldens, bench cfg3 A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -s -k "phase_scatter or subpixel or vae or pipeline" -x > gpurun_out/r2l_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "rel-err|PSNR" gpurun_out/r2l_pytest.log | tail -16; tail -4 gpurun_out/r2l_pytest.log
for f in "" "--two-step-upsampler" "" "--two-step-upsampler"; do
  timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline $f > gpurun_out/r2l_bench.json 2> gpurun_out/r2l_bench.err
  echo -n "bench [$f] rc=$? "; python - <<PY
import json
d=json.load(open('gpurun_out/r2l_bench.json'))
print({k:round(d[k],1) for k in ('ms_per_step','dit_ms_per_step','vae_encode_ms','vae_decode_ms','executed_tflop_per_step','achieved_tflops_per_gpu')}, d['roofline']['per_kernel'])
PY
done
