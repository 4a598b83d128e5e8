#!/bin/bash
# Next round: GroupNorm statistics fused into the sub-pixel conv kernel's epilogue (branch next/convsub-gnstats, never run) --
# kernel parity, the VAE goldens, then a cfg3 step (main: groupnorm_stats_kernel 82 ms per step, decode 5.45 s)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -s -x -k "fused_groupnorm or subpixel or vae" > gpurun_out/r3_gnstats_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "rel-err|PSNR|passed|failed" gpurun_out/r3_gnstats_pytest.log | tail -14
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_gnstats_bench.json 2> gpurun_out/r3_gnstats_bench.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/r3_gnstats_bench.json
