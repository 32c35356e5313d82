#!/bin/bash
# configs[4] driver, eager loop vs captured iteration on the same schedule -> gpurun_out/<tag>/train_synth_*.json
# usage: scripts/train_synth_pair.sh <tag> [--frames N --final G --iters-per-frame I --n-max-frames M ...]
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
for mode in graph eager; do
  extra=""; [ $mode = graph ] && extra="--graph"
  timeout 900 python scripts/train_synth.py "$@" $extra --json $O/train_synth_$mode.json > /dev/null 2> $O/train_synth_$mode.err
  echo "$mode rc=$?"
  python - <<PY
import json
d = json.load(open("$O/train_synth_$mode.json"))
print("$mode", {k: d[k] for k in ("iterations", "ms_per_iteration_by_resolution", "iterations_by_resolution", "graph", "loss_first", "loss_last", "fields", "frames", "peak_memory_GB")})
PY
done
