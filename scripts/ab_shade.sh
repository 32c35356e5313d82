#!/bin/bash
# Interleaved A/B of experiment builds (scripts/build_variant.sh) on one box: colour-kernel time of each, three rounds.
# usage: scripts/ab_shade.sh base pipe ...
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in "$@"; do
    rm -f $R/gpurun_out/diag.log
    LRF_LIB=$R/localrf_amd/csrc/liblrf_$v.so DIAG_STAGES=shade3_phases timeout 120 python -u $R/scripts/gpu_diag.py > /dev/null 2>&1
    echo "round $round $v: $(grep -E 'engine bf16x3' $R/gpurun_out/diag.log | cut -c1-120)"
    echo "         $(grep -E 'k_shade3 8' $R/gpurun_out/diag.log | cut -c20-200)"
  done
done
