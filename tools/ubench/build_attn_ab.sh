#!/bin/bash
# builds tools/ubench/attn_ab against the in-tree libseedvr2_hip.so (rpath relative to the binary, so it runs on the GPU box's copy)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 attn_ab.hip -o attn_ab \
    -L../../comfyui-seedvr2_videoupscaler_amd/csrc -lseedvr2_hip -Wl,-rpath,'$ORIGIN/../../comfyui-seedvr2_videoupscaler_amd/csrc'
echo built tools/ubench/attn_ab
