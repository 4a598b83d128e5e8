#!/bin/bash
# The experiments queued at the end of round 4, ready for ONE ~15 s gpurun call:
#   variant   (suffix _x): -DSVR_EP_ADDR=1                      conv epilogue addresses from column + wave-uniform row terms
#   variant2  (suffix _y): -DSVR_EP_ADDR=1 -DSVR_GN_TAIL_LDS=1 -DSVR_ACC_EARLY=1 -DSVR_GN_PACKED=1     ... + the other three switches
# Run here (build container) to build the product harnesses and the two experiment libraries; it prints the gpurun command.
# Every experiment is bit-identical by construction: equal `checksum` / `gn_checksum` columns are the acceptance test, the `us` columns the verdict.
set -e
cd "$(dirname "$0")"
bash build_ubench.sh
bash build_variant.sh -DSVR_EP_ADDR=1
SVR_VARIANT_DIR=variant2 SVR_VARIANT_SUFFIX=_y bash build_variant.sh -DSVR_EP_ADDR=1 -DSVR_GN_TAIL_LDS=1 -DSVR_ACC_EARLY=1 -DSVR_GN_PACKED=1
cat <<'CMD'

/usr/local/graft/bin/gpurun --timeout 60 -- 'mkdir -p gpurun_out; cd tools/ubench; C=c128,c256,c512,c128r,c256s,sp256,sp512,cin; G="gn128 gn256 gn512 gn128b";
 (for b in conv_ab conv_ab_x conv_ab_y conv_ab; do timeout 8 ./$b 5 $C; done; for b in gn_ab gn_ab_y gn_ab; do timeout 6 ./$b 10 $G; done; for cap in 4096 2048 1024; do timeout 6 ./gn_ab 10 gn128 gn256 gn_grid_cap=$cap; done;
  timeout 10 ./stream_ab 5 gn128) > ../../gpurun_out/queued_experiments.txt 2>&1; tail -n 80 ../../gpurun_out/queued_experiments.txt'
CMD
