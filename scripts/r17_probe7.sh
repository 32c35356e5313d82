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
print("eng $E", {k: round(v, 3) for k, v in d["ms_per_iteration_by_resolution"].items()}, "loss", d["loss_last"], "host: %.2f s inside replay(), waited %.2f (input ring) + %.2f (late reads) s of %.2f s" % (d["graph"].get("replay_host_s", -1), d["graph"].get("stage_wait_s", -1), d.get("late_reads_wait_s", -1), sum(d["ms_per_iteration_by_resolution"][k] * d["iterations_by_resolution"][k] for k in d["iterations_by_resolution"]) / 1e3))
print("   host ms per iteration:", {r: {k: round(x, 3) for k, x in v.items()} for r, v in d.get("host_ms_per_iteration_by_resolution", {}).items()})
print("   pace:", " ".join("%d:%.2f" % (a, b) for a, b in d.get("ms_per_iteration_by_250", [])))
PY
done
