#!/bin/bash
# round 6: density scatter through 3-D bricks (eng 1) against the per-(sample, plane) tile scatter (eng 17)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
cd $R
for G in ${GRIDS:-300 64 128 500 640}; do
  timeout 300 python scripts/bwd_probe.py --eng 1,17 --grid $G --steps 30 > $O/brick_vs_tiles_$G.txt 2>&1
  grep -v amdgpu.ids $O/brick_vs_tiles_$G.txt | cut -c1-200
done
bash scripts/serial_trace.sh fix TRAIN_ENG=1
