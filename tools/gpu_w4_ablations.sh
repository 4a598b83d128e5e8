# measurement build (SVR_BUILD_ABLATIONS=1 python -c 'import ...hip_lib as h; h.build(force=True)' beforehand): what each part of gemm_w4_kernel costs in place
for abl in 0 1 8 16 23; do echo "== pipe_abl=$abl"; SVR_BUILD_ABLATIONS=1 SVR_OPTIONS=gemm_w4=1,pipe_abl=$abl timeout 120 python tools/kbench.py --only gemm --reps 3 2>&1 | tail -6 | cut -c1-200; done
