#!/bin/bash
# Round-end evidence: kernel trace + PMC passes of bench.py, kernel trace of the backward, all
# summarised ON the GPU box into small markdown/json files (the rocpd databases stay there).
# usage: scripts/gpu_profile_round.sh <tag>
TAG=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python -u $R/bench.py --steps 50 --warmup 5 --no-baselines > $O/prof_bench.log 2>&1)
python $R/scripts/rocpd_stats.py $(find /tmp/prof_$TAG -name "*.db" | head -1) > $O/kernel_trace_fwd.md
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
            "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_${TAG}_$i -o pmc -- python -u $R/bench.py --steps 10 --warmup 2 --no-baselines > $O/pmc_$i.log 2>&1)
  echo "pmc pass $i rc=$?"
done
python $R/scripts/rocpd_pmc.py $(find /tmp/pmc_${TAG}_* -name "*.db") > $O/pmc_fwd.md
(cd /tmp && DIAG_STAGES=bwd timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_bwd_$TAG -o bwd -- python -u $R/scripts/gpu_diag.py > $O/prof_bwd.log 2>&1)
python $R/scripts/rocpd_stats.py $(find /tmp/prof_bwd_$TAG -name "*.db" | head -1) > $O/kernel_trace_bwd.md
grep -E "config2|grad rel" $R/gpurun_out/diag.log | tail -2 > $O/bwd_summary.txt
(cd /tmp && DIAG_STAGES=scene timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_scene_$TAG -o sc -- python -u $R/scripts/gpu_diag.py > $O/prof_scene.log 2>&1)
python $R/scripts/rocpd_busy.py $(find /tmp/prof_scene_$TAG -name "*.db" | head -1) 0.15 > $O/scene_iteration_busy.txt
DIAG_STAGES=scene,adam,reg timeout 300 python -u $R/scripts/gpu_diag.py > /dev/null 2>&1
grep -E "scene|Adam|repack|density_L1|TV_loss" $R/gpurun_out/diag.log | tail -12 > $O/scene_adam_reg.txt
ls -la $O
