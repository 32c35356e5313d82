#!/bin/bash
# round 6, first look: phase profile of the scatter kernels (prof builds) and the timeline of a captured 64^3 iteration
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
cd $R
LRF_LIB=$R/localrf_amd/csrc/liblrf_prof.so timeout 300 python scripts/scatter_prof_probe.py > $O/scatter_prof_base.txt 2>&1
LRF_LIB=$R/localrf_amd/csrc/liblrf_proft1.so timeout 300 python scripts/scatter_prof_probe.py > $O/scatter_prof_t1.txt 2>&1
LRF_LIB=$R/localrf_amd/csrc/liblrf_prof.so timeout 300 python scripts/scatter_prof_probe.py --grid 64 --samples 400 > $O/scatter_prof_base_64.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_g64 -o g -- python -u $R/scripts/graph_replay_probe.py --max-iters 300 > $O/prof_graph64.log 2>&1)
DB=$(find /tmp/prof_g64 -name "*.db" | head -1)
python scripts/rocpd_timeline.py $DB k_pack 3 > $O/graph_iteration_timeline_64.md
python scripts/rocpd_busy.py $DB 0.1 > $O/graph_busy_64.txt
tail -5 $O/prof_graph64.log
cat $O/scatter_prof_base.txt
