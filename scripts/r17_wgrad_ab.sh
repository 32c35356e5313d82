#!/bin/bash
# uncontended kernel times of the backward (one stream) for two engine settings: rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r17
mkdir -p $O
for E in ${ENGS:-1 513}; do
  (cd /tmp && rm -rf /tmp/prof_w$E && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w$E -o w -- python $R/scripts/bwd_probe.py --eng $E --overlap 0 --rounds 1 --steps 30 --grid ${GRID:-300} > /dev/null 2>&1)
  echo "== eng $E"
  python $R/scripts/rocpd_stats.py $(find /tmp/prof_w$E -name "*.db" | head -1) | cut -c1-150 | head -${TOP:-14}
done
