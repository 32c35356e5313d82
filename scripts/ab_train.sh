#!/bin/bash
# A/B of experiment builds (scripts/build_variant.sh) on the training step: gradient errors of the small golden, forward+backward
# time at configs[1], and the kernel table of a rocprofv3 trace of the same stage.  usage: scripts/ab_train.sh base variant ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  rm -f $R/gpurun_out/diag.log
  (cd /tmp && rm -rf /tmp/prof_ab_$v && LRF_LIB=$R/localrf_amd/csrc/liblrf_$v.so DIAG_STAGES=bwd timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_ab_$v -o t -- python -u $R/scripts/gpu_diag.py > /dev/null 2>&1)
  echo "== $v: $(grep -E 'config2 train' $R/gpurun_out/diag.log | cut -c1-120)"
  echo "   $(grep -E 'grad rel-to-max' $R/gpurun_out/diag.log | cut -c1-400)"
  python $R/scripts/rocpd_stats.py $(find /tmp/prof_ab_$v -name "*.db" | head -1) | grep -E "k_bin_fill|k_bwd_ray|k_train_app3|k_scatter|k_wgrad|k_train_dgrad|k_march|k_shade3" | cut -d'|' -f2-6
done
for round in 1 2; do for v in "$@"; do
  rm -f $R/gpurun_out/diag.log
  LRF_LIB=$R/localrf_amd/csrc/liblrf_$v.so DIAG_STAGES=bwd timeout 200 python -u $R/scripts/gpu_diag.py > /dev/null 2>&1
  echo "untraced round $round $v: $(grep -E 'config2 train' $R/gpurun_out/diag.log | cut -c1-120)"
done; done
