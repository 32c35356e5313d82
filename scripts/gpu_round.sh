#!/bin/bash
# One GPU visit: parity tests, bench line, rocprofv3 kernel-trace stats.  Outputs -> gpurun_out/.
# usage: scripts/gpu_round.sh <tag> [pytest|bench|prof ...]
TAG=${1:-r01}; shift
WHAT=${@:-pytest bench prof}
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in $WHAT; do
  case $w in
    pytest)
      timeout 420 python -u -m pytest tests -m gpu -q --timeout 120 -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log; tail -25 gpurun_out/pytest_$TAG.log ;;
    bench)
      timeout 300 python -u bench.py --steps 200 --warmup 20 > gpurun_out/bench_$TAG.log 2>&1
      echo "bench rc=$?" >> gpurun_out/bench_$TAG.log; tail -3 gpurun_out/bench_$TAG.log ;;
    prof)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- \
         python -u $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-baselines > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
      echo "prof rc=$?" >> gpurun_out/prof_$TAG.log
      find gpurun_out/prof_$TAG -name "*stats*" | head; f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && head -12 "$f" ;;
    pmc)
      # PMC passes, each in its own run with --kernel-trace only (no other trace domains)
      i=0
      for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
                  "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA TA_BUSY_avr GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i -o pmc -- \
           python -u $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-baselines > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i.log 2>&1)
        echo "pmc pass $i rc=$? ($ctrs)"; ls gpurun_out/pmc_${TAG}_$i 2>/dev/null | head -3
      done ;;
  esac
done
