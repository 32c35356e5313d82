#!/bin/bash
# One visit for a layout change of the saved rows: same-box A/B of library builds, the gradient tests, and a serial
# (one-stream) kernel trace of the backward.  usage: scripts/gpu_visit_frag.sh <tag> "<lib list for ab_libs>" "<pytest -k>" [variant the tests and the trace run with]
TAG=$1; LIBS=$2; KEXPR=$3; VAR=$4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
scripts/ab_libs.sh $LIBS > gpurun_out/ab_$TAG.log 2>&1; cat gpurun_out/ab_$TAG.log
unset LRF_LIB
if [ -n "$VAR" ]; then export LRF_LIB=$PWD/localrf_amd/csrc/liblrf_hip_$VAR.so; echo "tests + trace with $LRF_LIB"; fi
if [ -n "$KEXPR" ]; then
  timeout 900 python -u -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "$KEXPR" > gpurun_out/pytest_$TAG.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log; grep -E "passed|failed|FAILED|ERROR|rc=|Error|assert" gpurun_out/pytest_$TAG.log | tail -20
fi
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_ser_$TAG -o ser -- python -u $R/scripts/train_serial_probe.py > $R/gpurun_out/serial_$TAG.log 2>&1)
python $R/scripts/rocpd_stats.py $(find /tmp/prof_ser_$TAG -name "*.db" | head -1) > $R/gpurun_out/serial_trace_$TAG.md
head -16 $R/gpurun_out/serial_trace_$TAG.md | cut -c1-130
