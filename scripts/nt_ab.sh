#!/bin/bash
# A/B of the non-temporal hint on the saved rows (LRF_ROW_NT): serial kernel times of the row producers / consumers
export TMPDIR=/tmp
for v in nt63 nt20 nt0 nt63 nt20 nt0; do
  LRF_LIB=$GRAFT_REPO_ROOT/localrf_amd/csrc/liblrf_$v.so bash scripts/serial_trace.sh $v > /dev/null 2>&1
  echo "== $v: $(grep 'fwd+bwd' gpurun_out/serial_$v.log)"
  grep -E "k_train_dgrad3|k_bwd_shade_fwd|k_wgrad|k_scatter_plane<24" gpurun_out/serial_$v.md | cut -d'|' -f2,5,6 | cut -c1-40,75-120
done
