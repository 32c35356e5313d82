#!/bin/bash
# timeline of one captured iteration at 500^3 late in the progressive schedule
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
cd $R
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_g500 -o g -- python -u $R/scripts/graph_replay_probe.py --final 500 --max-iters ${ITERS:-7600} > $O/prof_graph500.log 2>&1)
tail -3 $O/prof_graph500.log | cut -c1-300
DB=$(find /tmp/prof_g500 -name "*.db" | head -1)
python scripts/rocpd_timeline.py $DB k_pack 3 > $O/graph_iteration_timeline_500.md
python scripts/rocpd_busy.py $DB 0.05 > $O/graph_busy_500.txt
cat $O/graph_iteration_timeline_500.md | cut -c1-110
head -16 $O/graph_busy_500.txt
