#!/bin/bash
# One gpurun call: kernel micro-bench (HIP-event timings) + rocprofv3 PMC passes over the same script.
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip traces).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ONLY=${1:-conv,gemm,attn,side}
TAG=${2:-r1}
timeout 600 python tools/kbench.py --reps 5 --only $ONLY > gpurun_out/kbench_$TAG.jsonl 2> gpurun_out/kbench_$TAG.err
echo "kbench rc=$?"; cat gpurun_out/kbench_$TAG.jsonl; tail -3 gpurun_out/kbench_$TAG.err
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${TAG}_$i
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace -d gpurun_out/pmc_${TAG}_$i -o pmc --output-format csv -- \
      python tools/kbench.py --reps 2 --only $ONLY > /dev/null 2> gpurun_out/pmc_${TAG}_$i.err
  echo "pmc pass $i ($PMC) rc=$?"; tail -2 gpurun_out/pmc_${TAG}_$i.err
done
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_ > gpurun_out/pmc_${TAG}_summary.txt 2>&1
cat gpurun_out/pmc_${TAG}_summary.txt | cut -c1-260
