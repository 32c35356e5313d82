#!/bin/bash
# One short visit to a GPU box: its identity, the exact-integer MFMA source-timing test (scripts/ubench/mfma_war), and
# the finding-17 screen (scripts/gpu_diag.py screen).  Outputs -> gpurun_out/screen_<tag>.log
TAG=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/screen_$TAG.log
{
  rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1
  (cd scripts/ubench && timeout 120 ./mfma_war ${MFMA_LAUNCHES:-100} 1500 | tail -7)
  rm -f gpurun_out/diag.log
  SCREEN_ROUNDS=${SCREEN_ROUNDS:-1500} DIAG_STAGES=screen timeout 400 python -u scripts/gpu_diag.py > /dev/null 2>&1
  grep -i "screen\|error\|Traceback\|rc " gpurun_out/diag.log | tail -14
} > $OUT 2>&1
cat $OUT
