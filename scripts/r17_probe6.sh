#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
cd $R
timeout 600 python -u scripts/graph_replay_probe.py --max-iters 300 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
timeout 600 python -u scripts/graph_replay_probe.py --max-iters 300 --profile 2>&1 | grep -v amdgpu.ids | tail -45 | cut -c1-200 | tee $O/graph_host_profile_64.txt
