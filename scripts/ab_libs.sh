#!/bin/bash
# Same-box A/B of library builds (scripts/build_variant.py): usage scripts/ab_libs.sh default ilp clause default ...
# prints the two-launch forward time and forward+backward of each.
for v in "$@"; do
  if [ "$v" = default ]; then unset LRF_LIB; else export LRF_LIB=$PWD/localrf_amd/csrc/liblrf_hip_$v.so; fi
  DIAG_STAGES=fuse,bwd_overlap timeout 300 python -u scripts/gpu_diag.py > /dev/null 2>&1
  f=$(grep "shade_pipe 0" gpurun_out/diag.log | tail -1 | sed 's/.*partials): \([0-9.]*\) ms.*colour \([0-9.]*\) fin.*/\1 ms (colour \2 us)/')
  b=$(grep "overlap 1.*row-saving forward k_bwd_shade_fwd, dW2 on split" gpurun_out/diag.log | tail -1 | sed 's/.*fwd+bwd \([0-9.]*\) ms (forward alone \([0-9.]*\) ms).*/\1 ms (train fwd \2)/')
  d=$(grep "renders differing" gpurun_out/diag.log | tail -1 | sed 's/.*first: //')
  echo "$v: forward $f | fwd+bwd $b | differing renders $d"
done
