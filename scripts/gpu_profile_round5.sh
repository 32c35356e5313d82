#!/bin/bash
# Round-5 evidence on ONE box: everything of scripts/gpu_profile_round4.sh (kernel trace of the bench command, PMC passes of the
# forward, uncontended kernel times + PMC of the training kernels, the full default bench line), then configs[4]: the
# progressive driver on the reference's schedule, eager loop and captured iteration, and a kernel trace of captured iterations
# (timeline of one iteration, GPU-busy fraction).  usage: scripts/gpu_profile_round5.sh <tag>
TAG=${1:-r15}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
bash $R/scripts/gpu_profile_round4.sh $TAG > /dev/null 2>&1
cd $R
scripts/train_synth_pair.sh $TAG --frames 16 --final 500 --iters-per-frame 600 --n-max-frames 12 > $O/train_synth_pair.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_g_$TAG -o g -- python -u $R/scripts/graph_replay_probe.py --max-iters 1800 > $O/prof_graph.log 2>&1)
DB=$(find /tmp/prof_g_$TAG -name "*.db" | head -1)
python scripts/rocpd_timeline.py $DB k_pack 3 > $O/graph_iteration_timeline.md
python scripts/rocpd_busy.py $DB 0.1 > $O/graph_busy.txt
ls -la $O
