#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for E in 1 257 513 769 1025 2049 3841; do
  echo "== TRAIN_ENG=$E"
  bash scripts/serial_trace.sh b$E TRAIN_ENG=$E | grep -E "k_scatter_brick|k_brick"
done
