#!/bin/bash
# builds an EXPERIMENT library next to the product one: tools/ubench/variant/libseedvr2_hip.so = the same sources compiled with the
# extra -D defines given on the command line (e.g. -DSVR_EP_DBUF=1), and tools/ubench/conv_ab_x / attn_ab_x / gemm_ab_x / gn_ab_x = the harnesses linked
# against it.  One gpurun call can then A/B product and experiment back to back:  conv_ab 5 c128 ; conv_ab_x 5 c128
# A second experiment next to the first:  SVR_VARIANT_DIR=variant2 SVR_VARIANT_SUFFIX=_y tools/ubench/build_variant.sh -D...
set -e
cd "$(dirname "$0")"
DIR=${SVR_VARIANT_DIR:-variant}
SUF=${SVR_VARIANT_SUFFIX:-_x}
mkdir -p "$DIR"
FLAGS=$(python3 -c "import importlib,sys; sys.path.insert(0,'../..'); h=importlib.import_module('comfyui-seedvr2_videoupscaler_amd.hip_lib'); print(' '.join(h.HIPCC_FLAGS))")
/opt/rocm/bin/hipcc $FLAGS "$@" -DSVR_BUILD_ID="\"variant $*\"" ../../comfyui-seedvr2_videoupscaler_amd/csrc/svr_api.hip -o "$DIR/libseedvr2_hip.so"
for t in attn_ab conv_ab gemm_ab gn_ab; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 $t.hip -o ${t}${SUF} -L"$DIR" -lseedvr2_hip -Wl,-rpath,"\$ORIGIN/$DIR" 2>/dev/null
done
echo built tools/ubench/$DIR/libseedvr2_hip.so "$@"
