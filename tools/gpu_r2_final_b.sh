#!/bin/bash
# Round 2 final measurements, part B: PMC passes -- HBM traffic of the cfg3 bench's kernels; MFMA-busy / clock of the conv kernel (8 vs 4 rows)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_pmc_bench.sh cfg3 r2 > gpurun_out/r2_pmcb.log 2>&1; tail -30 gpurun_out/r2_pmcb.log | cut -c1-180
SVR_OPTIONS=conv_rows=8 bash tools/gpu_pmc_conv.sh rows8 "--match 128->128" > gpurun_out/r2_pmc_rows8.log 2>&1; tail -25 gpurun_out/r2_pmc_rows8.log | cut -c1-180
SVR_OPTIONS=conv_rows=4 bash tools/gpu_pmc_conv.sh rows4 "--match 128->128" > gpurun_out/r2_pmc_rows4.log 2>&1; tail -25 gpurun_out/r2_pmc_rows4.log | cut -c1-180
find gpurun_out -name "*counter_collection.csv" -delete; find gpurun_out -name "*kernel_trace.csv" -delete
