#!/bin/bash
# One GPU visit of round 2: parity tests, hazard experiment, bench line.  Outputs -> gpurun_out/.
# usage: scripts/gpu_round2.sh <tag> [pytest|diag|bench|prof|pmc ...]
TAG=${1:-r02a}; shift
WHAT=${@:-pytest diag bench}
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in $WHAT; do
  case $w in
    pytest)
      timeout 900 python -u -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -s > gpurun_out/pytest_$TAG.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log; grep -E "passed|failed|FAILED|ERROR|^fuzz|^flips|rc=" gpurun_out/pytest_$TAG.log | tail -40 ;;
    diag)
      DIAG_STAGES=${DIAG_STAGES:-mfma_policy} timeout 600 python -u scripts/gpu_diag.py > gpurun_out/diag_$TAG.log 2>&1
      echo "diag rc=$?" >> gpurun_out/diag_$TAG.log; tail -15 gpurun_out/diag_$TAG.log ;;
    bench)
      timeout 600 python -u bench.py --steps 200 --warmup 20 > gpurun_out/bench_$TAG.log 2>&1
      echo "bench rc=$?" >> gpurun_out/bench_$TAG.log; tail -c 6000 gpurun_out/bench_$TAG.log ;;
    prof)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- \
         python -u $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-baselines --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
      echo "prof rc=$?" >> gpurun_out/prof_$TAG.log
      python scripts/rocpd_stats.py $(find /tmp/prof_$TAG -name "*.db" | head -1) > gpurun_out/kernel_trace_$TAG.md 2>&1
      head -30 gpurun_out/kernel_trace_$TAG.md ;;
    pmc)
      i=0
      for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
                  "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA TA_BUSY_avr GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_${TAG}_$i -o pmc -- \
           python -u $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --child > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i.log 2>&1)
        echo "pmc pass $i rc=$?"
      done
      python scripts/rocpd_pmc.py $(find /tmp/pmc_${TAG}_* -name "*.db") > gpurun_out/pmc_$TAG.md 2>&1
      cat gpurun_out/pmc_$TAG.md ;;
  esac
done
