#!/bin/bash
# ONE parametrised gpurun launcher (replaces the per-experiment gpu_r2?.sh scripts).  usage, as one gpurun call:
#   gpurun --timeout S -- 'bash tools/gpu.sh <tag> <step> [<step> ...]'
# steps (each writes gpurun_out/<tag>_<step>.log and prints a short tail):
#   tests[:<-k expr>]    pytest -m gpu (whole suite, or the -k selection)
#   variants             pytest -m variants (non-default kernel generations)
#   smoke                __graft_entry__.smoke()
#   kbench[:<only>[:<SVR_OPTIONS>]]   tools/kbench.py --only <only>
#   bench[:<workload>[:<extra args, '+' for spaces>]]   bench.py --steps 2 --warmup 1 (no cpu baseline)
#   benchfull[:<workload>]  bench.py with its defaults (the driver's line)
#   prof[:<workload>]    rocprofv3 --kernel-trace --stats of a 2-step bench -> gpurun_out/<tag>_prof_kernel_stats.csv
#   pmc[:<workload>]     separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA busy / L2) of a 1-step bench
#   trace[:<only>]       rocprofv3 --kernel-trace --stats of tools/kbench.py --only <only>
#   py:<script>[:args+with+plus]     python <script> args
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=$1; shift
for step in "$@"; do
  IFS=':' read -r kind a b c <<< "$step"
  log=gpurun_out/${TAG}_$(echo "$step" | tr -c 'A-Za-z0-9_.\n' '_').log
  t0=$(date +%s)
  case $kind in
    tests)    timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=6 ${a:+-k "$a"} -s > $log 2>&1; rc=$?
              grep -E "rel-err|PSNR|dB|passed|failed|Error" $log | tail -${TAILN:-12} ;;
    variants) timeout 900 python -m pytest tests -q -m variants -p no:cacheprovider > $log 2>&1; rc=$?; tail -4 $log ;;
    smoke)    timeout 600 python __graft_entry__.py smoke > $log 2>&1; rc=$?; tail -5 $log ;;
    kbench)   SVR_OPTIONS=${b:-} timeout 600 python tools/kbench.py --reps 5 --only ${a:-conv,gemm,attn,side} > $log 2> $log.err; rc=$?; cat $log ;;
    bench)    timeout 1500 python bench.py --workload ${a:-cfg3} --steps 2 --warmup 1 --no-cpu-baseline ${b//+/ } > $log 2> $log.err; rc=$?
              cut -c1-1500 $log; tail -2 $log.err ;;
    benchfull) timeout 1500 python bench.py --workload ${a:-cfg3} > $log 2> $log.err; rc=$?; cat $log; tail -2 $log.err ;;
    prof)     d=gpurun_out/${TAG}_prof; rm -rf $d
              timeout 1500 rocprofv3 --kernel-trace --stats -d $d -o prof --output-format csv -- python bench.py --workload ${a:-cfg3} --steps 2 --warmup 1 --no-cpu-baseline > $log 2>&1; rc=$?
              f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_prof_kernel_stats.csv && head -25 $f
              tr=$(find $d -name '*kernel_trace.csv' | head -1); [ -n "$tr" ] && python tools/trace_by_shape.py $tr > gpurun_out/${TAG}_prof_by_shape.txt 2>/dev/null
              rm -rf $d ;;
    pmc)      rc=0; i=0
              for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
                i=$((i+1)); d=gpurun_out/${TAG}_pmc_$i; rm -rf $d
                timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d $d -o pmc --output-format csv -- python bench.py --workload ${a:-cfg3} --steps 1 --warmup 0 --no-cpu-baseline > $log.$i 2>&1 || rc=$?
              done
              python tools/pmc_summary.py gpurun_out/${TAG}_pmc_ > gpurun_out/${TAG}_pmc_summary.txt 2>&1
              python tools/pmc_traffic_json.py gpurun_out/${TAG}_pmc_ ${a:-cfg3} > gpurun_out/${TAG}_pmc_traffic.json 2>> $log.1
              for j in 1 2 3 4; do rm -rf gpurun_out/${TAG}_pmc_$j; done
              head -30 gpurun_out/${TAG}_pmc_summary.txt ;;
    trace)    d=gpurun_out/${TAG}_trace; rm -rf $d      # kernel names + per-kernel time of a kbench selection (e.g. the vendor GEMM's kernel)
              timeout 900 rocprofv3 --kernel-trace --stats -d $d -o t --output-format csv -- python tools/kbench.py --reps 3 --only ${a:-gemm} > $log 2>&1; rc=$?
              f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_trace_${a:-gemm}_kernel_stats.csv && cut -c1-400 $f | head -12
              rm -rf $d ;;
    py)       timeout 1500 python $a ${b//+/ } > $log 2>&1; rc=$?; tail -${TAILN:-25} $log ;;
    *)        echo "unknown step $step"; rc=99 ;;
  esac
  echo "== $step rc=$rc $(( $(date +%s) - t0 ))s"
done
