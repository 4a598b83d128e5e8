#!/bin/bash
# L2 behaviour of the conv kernel at a power-of-two vs a non-power-of-two row pitch (separate --pmc passes)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for SH in "128->128 @1024^2" "128->128 @1024x1056"; do
  TAG=$(echo "$SH" | tr -c 'a-zA-Z0-9' '_')
  i=0
  for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    rm -rf gpurun_out/pmcl2_${TAG}_$i
    timeout 200 rocprofv3 --pmc $PMC --kernel-trace -d gpurun_out/pmcl2_${TAG}_$i -o pmc --output-format csv -- \
        python tools/kbench.py --reps 2 --only conv --match "$SH" > /dev/null 2> gpurun_out/pmcl2_${TAG}_$i.err
    echo "pass $i rc=$?"
  done
  echo "=== $SH"
  python tools/pmc_summary.py gpurun_out/pmcl2_${TAG}_ 2>&1 | grep -A14 "conv_halo2" | cut -c1-150
done
