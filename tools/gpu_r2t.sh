#!/bin/bash
# Round 2: where does cfg4 lose time with the split causal head?  kernel + HIP API stats of one clip
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
rm -rf gpurun_out/prof_r2t
timeout 500 rocprofv3 --kernel-trace --hip-runtime-trace --stats -d gpurun_out/prof_r2t -o prof --output-format csv -- \
    python bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline --three-tap-head > gpurun_out/r2t_prof.json 2> gpurun_out/r2t_prof.err
echo "prof rc=$?"; cut -c1-200 gpurun_out/r2t_prof.json; tail -3 gpurun_out/r2t_prof.err | cut -c1-200
find gpurun_out/prof_r2t -name "*_trace.csv" -delete
ls gpurun_out/prof_r2t
for f in $(find gpurun_out/prof_r2t -name "*stats.csv"); do echo "== $f"; head -14 "$f" | cut -c1-160; done
