#!/bin/bash
# Round 2: two-term causal head + thin-output conv kernel -- full GPU suite, smoke, bench cfg3 (final line), head A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=4 -s > gpurun_out/r2p_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "rel-err|PSNR" gpurun_out/r2p_pytest_gpu.log | tail -26; tail -8 gpurun_out/r2p_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r2p_smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/r2p_smoke.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2p_bench_cfg3.json 2> gpurun_out/r2p_bench_cfg3.err
echo "bench cfg3 rc=$?"; cut -c1-400 gpurun_out/r2p_bench_cfg3.json; tail -2 gpurun_out/r2p_bench_cfg3.err | cut -c1-300
i=0
for f in "--three-tap-head" "" ; do
  i=$((i+1))
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $f > gpurun_out/r2p_ab_$i.json 2> gpurun_out/r2p_ab_$i.err
  echo -n "bench [$f] rc=$? "; python - <<PY
import json
d=json.load(open('gpurun_out/r2p_ab_$i.json'))
print({k:round(d[k],1) for k in ('ms_per_step','dit_ms_per_step','vae_encode_ms','vae_decode_ms','executed_tflop_per_step','achieved_tflops_per_gpu')}, d['roofline']['per_kernel'])
PY
done
rm -rf gpurun_out/prof_r2p
timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2p -o prof --output-format csv -- \
    python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2p_prof.json 2> gpurun_out/r2p_prof.err
echo "prof rc=$?"; cut -c1-200 gpurun_out/r2p_prof.json
t=$(find gpurun_out/prof_r2p -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_by_shape.py "$t" > gpurun_out/r2p_kernels_by_shape.txt
find gpurun_out/prof_r2p -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/prof_r2p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220
