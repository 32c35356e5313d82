#!/bin/bash
# One short GPU visit: selected parity tests, a diag stage list, bench lines at the driver's K/W.  Outputs -> gpurun_out/.
# usage: scripts/gpu_visit.sh <tag> "<pytest -k expression or empty>" "<diag stages or empty>" [bench]
TAG=$1; KEXPR=$2; STAGES=$3; BENCH=$4
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  timeout 900 python -u -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -s -k "$KEXPR" > gpurun_out/pytest_$TAG.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log; grep -E "passed|failed|FAILED|ERROR|rc=|Error" gpurun_out/pytest_$TAG.log | tail -20
fi
if [ -n "$STAGES" ]; then
  DIAG_STAGES=$STAGES timeout 900 python -u scripts/gpu_diag.py > gpurun_out/diag_$TAG.log 2>&1
  echo "diag rc=$?" >> gpurun_out/diag_$TAG.log; tail -25 gpurun_out/diag_$TAG.log
fi
if [ -n "$BENCH" ]; then
  for i in 1 2 3; do
    timeout 300 python -u bench.py --gpus 1 --steps 20 --warmup 5 --no-baselines --no-pmc 2>/dev/null | python -c "import sys,json; [print('K20', json.loads(l)['value'], json.loads(l)['ms_per_step'], json.loads(l)['train_step']['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
  timeout 300 python -u bench.py --gpus 1 --steps 20 --warmup 5 --no-baselines --no-pmc --preroll-ms 1500 2>/dev/null | python -c "import sys,json; [print('K20 preroll1500', json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  timeout 300 python -u bench.py --gpus 1 --steps 200 --warmup 20 --no-baselines --no-pmc 2>/dev/null | python -c "import sys,json; [print('K200', json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
fi
