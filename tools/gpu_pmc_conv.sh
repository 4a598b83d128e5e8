#!/bin/bash
# conv kernels only: MFMA-busy fraction, shader clock under load, LDS / VMEM activity (separate --pmc passes)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-conv}; EXTRA=${2:-}
i=0
for PMC in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${TAG}_$i
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace -d gpurun_out/pmc_${TAG}_$i -o pmc --output-format csv -- \
      python tools/kbench.py --reps 2 --only conv $EXTRA > /dev/null 2> gpurun_out/pmc_${TAG}_$i.err
  echo "pmc pass $i rc=$?"; tail -1 gpurun_out/pmc_${TAG}_$i.err
done
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_ > gpurun_out/pmc_${TAG}_summary.txt 2>&1
grep -A40 "conv_halo2" gpurun_out/pmc_${TAG}_summary.txt | cut -c1-200 | head -70
