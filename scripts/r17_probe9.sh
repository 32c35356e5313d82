#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
echo "== default"; bash scripts/serial_trace.sh d$i TRAIN_ENG=1 | grep -E "k_train_app3|k_bwd_ray"
echo "== novmax"; bash scripts/serial_trace.sh n$i TRAIN_ENG=1 LRF_LIB=$R/localrf_amd/csrc/liblrf_novmax.so | grep -E "k_train_app3|k_bwd_ray"
done
