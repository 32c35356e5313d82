#!/bin/bash
# Round-4 evidence on ONE box: (1) kernel trace of the bench command (forward + train_step), (2) PMC passes of the forward,
# (3) uncontended kernel times + PMC of the training kernels (both branches of the backward on one stream), (4) the full
# default bench line.  Summaries are written on the box; the rocpd databases stay there.
# usage: scripts/gpu_profile_round4.sh <tag>
TAG=${1:-r13}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python -u $R/bench.py --steps 50 --warmup 5 --no-baselines --no-pmc > $O/prof_bench.log 2>&1)
python $R/scripts/rocpd_stats.py $(find /tmp/prof_$TAG -name "*.db" | head -1) > $O/kernel_trace_bench.md
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
            "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_${TAG}_$i -o pmc -- python -u $R/bench.py --steps 10 --warmup 2 --child fwd > $O/pmc_$i.log 2>&1)
  echo "pmc pass $i rc=$?"
done
python $R/scripts/rocpd_pmc.py $(find /tmp/pmc_${TAG}_* -name "*.db") > $O/pmc_fwd.md
bash $R/scripts/serial_trace.sh $TAG > /dev/null 2>&1
cp $R/gpurun_out/serial_$TAG.md $O/kernel_trace_train_serial.md; grep "fwd+bwd" $R/gpurun_out/serial_$TAG.log > $O/fwd_bwd_ms.txt
bash $R/scripts/gpu_pmc_train.sh $TAG > /dev/null 2>&1
cp $R/gpurun_out/pmc_train_$TAG.md $O/pmc_train.md
(time timeout 600 python -u $R/bench.py) > $O/bench_default.log 2>&1
grep '^{' $O/bench_default.log > $O/bench_default.json
ls -la $O
