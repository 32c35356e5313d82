#!/bin/bash
# round 6: fixed-point scatters (eng 1: both, 257: density only) against the compare-and-swap scatters (eng 17)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
cd $R
for G in ${GRIDS:-300 64 128 500 640}; do
  timeout 300 python scripts/bwd_probe.py --eng ${ENGS:-1,257,17} --grid $G --steps 30 > $O/fix_vs_cas_$G.txt 2>&1
  grep -v amdgpu.ids $O/fix_vs_cas_$G.txt | cut -c1-200
done
bash scripts/serial_trace.sh fix TRAIN_ENG=1
