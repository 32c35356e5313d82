#!/bin/bash
# Uncontended durations of the backward kernels: scripts/train_serial_probe.py under rocprofv3 --kernel-trace.
# usage: scripts/serial_trace.sh <tag> [env assignments...]  -> gpurun_out/serial_<tag>.md
TAG=${1:-x}; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(cd /tmp && env "$@" timeout 200 rocprofv3 --kernel-trace -d /tmp/ser_$TAG -o ser -- python -u $R/scripts/train_serial_probe.py > $R/gpurun_out/serial_$TAG.log 2>&1)
python $R/scripts/rocpd_stats.py $(find /tmp/ser_$TAG -name "*.db" | head -1) > $R/gpurun_out/serial_$TAG.md
grep -E "lrf::" $R/gpurun_out/serial_$TAG.md | cut -c1-60,75-130 | head -24
