#!/bin/bash
# HBM traffic of the bench workload's kernels: two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE need
# separate passes) over one un-warmed bench step.  Only --kernel-trace is combined with --pmc.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WL=${1:-cfg3}; TAG=${2:-r1}
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcb_${TAG}_$i
  timeout 900 rocprofv3 --pmc $PMC --kernel-trace -d gpurun_out/pmcb_${TAG}_$i -o pmc --output-format csv -- \
      python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmcb_${TAG}_$i.json 2> gpurun_out/pmcb_${TAG}_$i.err
  echo "pmc pass $i ($PMC) rc=$?"; tail -2 gpurun_out/pmcb_${TAG}_$i.err | cut -c1-300
  find gpurun_out/pmcb_${TAG}_$i -name "*kernel_trace.csv" -delete
done
python tools/pmc_summary.py gpurun_out/pmcb_${TAG}_ > gpurun_out/pmcb_${TAG}_summary.txt 2>&1
python tools/pmc_traffic_json.py gpurun_out/pmcb_${TAG}_ $WL > gpurun_out/pmcb_${TAG}_traffic.json
find gpurun_out -name "*counter_collection.csv" -path "*pmcb_${TAG}_*" -delete
grep -A12 "conv_halo2" gpurun_out/pmcb_${TAG}_summary.txt | cut -c1-160 | head -40
