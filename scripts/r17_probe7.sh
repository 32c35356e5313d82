#!/bin/bash
# configs[4] on the reference schedule, captured iterations only (LRF_ENG: lrf_debug_set_train_fwd_engine bits for an A/B)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
cd $R
for E in ${ENGS:-1}; do
  LRF_TRAIN_ENG=$E timeout 900 python scripts/train_synth.py --frames 16 --final 500 --iters-per-frame 600 --n-max-frames 12 --graph --json $O/train_graph_eng$E.json > /dev/null 2> $O/train_graph_eng$E.err
  python - <<PY
import json
d = json.load(open("$O/train_graph_eng$E.json"))
print("eng $E", {k: round(v, 3) for k, v in d["ms_per_iteration_by_resolution"].items()}, "loss", d["loss_last"])
PY
done
