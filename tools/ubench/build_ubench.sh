#!/bin/bash
# builds the standalone C-ABI harnesses (attn_ab, conv_ab, gemm_ab, gn_ab) against the in-tree libseedvr2_hip.so (rpath relative to the binary, so
# they run from the GPU box's copy of the tree); the binaries are git-ignored and travel with gpurun like the library itself
set -e
cd "$(dirname "$0")"
for t in attn_ab conv_ab gemm_ab gn_ab; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 $t.hip -o $t \
        -L../../comfyui-seedvr2_videoupscaler_amd/csrc -lseedvr2_hip -Wl,-rpath,'$ORIGIN/../../comfyui-seedvr2_videoupscaler_amd/csrc'
    echo built tools/ubench/$t
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 stream_ab.hip -o stream_ab && echo built tools/ubench/stream_ab
