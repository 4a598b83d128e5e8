#!/bin/bash
# Round 2: window-attention kernel -- parity, A/B of the build variants, PMC passes (separate runs, --kernel-trace only).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
# bail out early on a sick box (seen once: every process aborted at its first allocation, 10 GPU-minutes burnt)
timeout 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); torch.cuda.synchronize(); assert float(x.sum())==1<<20" || { echo 'GPU sanity check failed'; exit 9; }
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -x -k "attn" > gpurun_out/r2b_kernels.log 2>&1
echo "kernels rc=$?"; tail -8 gpurun_out/r2b_kernels.log
timeout 400 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -s -k "multiwindow" > gpurun_out/r2b_parity.log 2>&1
echo "parity rc=$?"; grep -E "rel-err|passed|failed|Error|error" gpurun_out/r2b_parity.log | tail -8
timeout 300 python tools/kbench.py --reps 5 --only attn --no-vae-attn > gpurun_out/r2b_kbench_attn.jsonl 2> gpurun_out/r2b_kbench_attn.err
echo "kbench rc=$?"; cat gpurun_out/r2b_kbench_attn.jsonl; tail -3 gpurun_out/r2b_kbench_attn.err
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_r2b_$i
  timeout 240 rocprofv3 --pmc $PMC --kernel-trace -d gpurun_out/pmc_r2b_$i -o pmc --output-format csv -- \
      python tools/kbench.py --reps 2 --only attn --no-vae-attn > /dev/null 2> gpurun_out/pmc_r2b_$i.err
  echo "pmc pass $i rc=$?"; tail -1 gpurun_out/pmc_r2b_$i.err
done
python tools/pmc_summary.py gpurun_out/pmc_r2b_ > gpurun_out/pmc_r2b_summary.txt 2>&1
cut -c1-200 gpurun_out/pmc_r2b_summary.txt
