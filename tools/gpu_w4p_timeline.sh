# measurement build (SVR_BUILD_ABLATIONS=1 python -c '...hip_lib.build(force=True)' beforehand): where gemm_w4p_kernel's workgroups spend a
# tile (K loop / epilogue issue / store drain, 100 MHz stamps) and how the XCDs' epilogues line up in time.
# pipe_abl 100 full | 101 no global stores | 108 no K-loop barrier (102 / 104 / 116 / 132 of profiles/r3_gemm_w4_ablations.txt: re-add their cases in launch_gemm_w4)
for abl in ${ABLS:-100 101 108}; do echo "== pipe_abl=$abl"; SVR_BUILD_ABLATIONS=1 SVR_OPTIONS=gemm_w4=2,pipe_abl=$abl timeout 300 python tools/kbench.py --only gemm --reps 1 2>&1 | grep -v amdgpu.ids | grep "tile 20\|kernel\|shader cycles" | cut -c1-200; done
