#!/bin/bash
# round 6: captured iteration at 64^3 in the regularised phase after the loss-assembly kernels and the L1 backward fix
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "assembly or geometric or density_l1 or L1 or captured or progressive" 2>&1 | tail -5
timeout 600 python -u scripts/graph_replay_probe.py --max-iters 300 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-600 | tee $O/graph_replay_64_after.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_g64 -o g -- python -u $R/scripts/graph_replay_probe.py --max-iters 300 > $O/prof_graph64_after.log 2>&1)
DB=$(find /tmp/prof_g64 -name "*.db" | head -1)
python scripts/rocpd_timeline.py $DB k_pack 3 > $O/graph_iteration_timeline_64_after.md
python scripts/rocpd_busy.py $DB 0.1 > $O/graph_busy_64_after.txt
head -3 $O/graph_iteration_timeline_64_after.md; head -18 $O/graph_busy_64_after.txt
