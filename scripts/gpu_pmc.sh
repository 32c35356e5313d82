#!/bin/bash
# SQ / TA counters of the forward kernels (two rocprofv3 --pmc passes over a short bench.py run), summarised on the box.
# usage: scripts/gpu_pmc.sh <tag>   -> gpurun_out/pmc_<tag>.md
TAG=${1:-x}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_${TAG}_$i -o pmc -- python -u $R/bench.py --steps 10 --warmup 2 --child > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1)
  echo "pmc pass $i rc=$?"
done
python $R/scripts/rocpd_pmc.py $(find /tmp/pmc_${TAG}_* -name "*.db") > $R/gpurun_out/pmc_$TAG.md
cat $R/gpurun_out/pmc_$TAG.md
