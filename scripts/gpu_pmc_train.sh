#!/bin/bash
# SQ counters of the training-step kernels (two rocprofv3 --pmc passes over scripts/train_serial_probe.py: both branches of
# the backward on ONE stream, so every kernel runs uncontended), summarised on the box.
# usage: scripts/gpu_pmc_train.sh <tag>   -> gpurun_out/pmc_train_<tag>.md
TAG=${1:-x}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && OVL2=0 timeout 200 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmct_${TAG}_$i -o pmc -- python -u $R/scripts/train_serial_probe.py > $R/gpurun_out/pmc_train_${TAG}_$i.log 2>&1)
  echo "pmc pass $i rc=$?"
done
python $R/scripts/rocpd_pmc.py $(find /tmp/pmct_${TAG}_* -name "*.db") > $R/gpurun_out/pmc_train_$TAG.md
cat $R/gpurun_out/pmc_train_$TAG.md
