#!/bin/bash
# round 6: the premultiplied-row appearance scatter (k_scatter_pre, integer LDS adds) against the dX-row scatter
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
cd $R
timeout 300 python scripts/bwd_probe.py --eng 1,17 > $O/pre_vs_rows.txt 2>&1
cat $O/pre_vs_rows.txt
bash scripts/serial_trace.sh pre TRAIN_ENG=1
bash scripts/serial_trace.sh rows TRAIN_ENG=17
