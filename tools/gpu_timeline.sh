#!/bin/bash
# s_memtime tile anatomy of the conv kernel (measurement build libseedvr2_hip_abl.so, built beforehand with SVR_BUILD_ABLATIONS=1)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SVR_BUILD_ABLATIONS=1
for shape in ${SHAPES:-"5,1024,1024,128,128" "9,256,256,512,512"}; do
  for opt in ${OPTS:-"conv_rows=8" "conv_rows=4"}; do
    echo "== SHAPE=$shape $opt"
    rows=8; case $opt in *conv_rows=4*) rows=4;; esac
    SVR_OPTIONS=$opt SHAPE=$shape CONV_ROWS=$rows timeout 300 python tools/conv_timeline.py 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/conv_timeline_r2.txt 2>&1
cat gpurun_out/conv_timeline_r2.txt
