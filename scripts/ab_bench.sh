#!/bin/bash
# Interleaved A/B of experiment builds (scripts/build_variant.sh) on one box: headline bench line of each, N rounds.
# usage: scripts/ab_bench.sh base pk ...
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in "$@"; do
    LRF_LIB=$R/localrf_amd/csrc/liblrf_$v.so timeout 300 python -u $R/bench.py --gpus 1 --steps 50 --warmup 10 --no-baselines --no-pmc 2>/dev/null | python -c "import sys,json; [print('round $round $v', round(json.loads(l)['value']/1e6,3), 'M rays/s', round(json.loads(l)['ms_per_step'],5), 'k_shade3', round(json.loads(l)['roofline']['kernels']['k_shade3']['ms'],5), 'train', round(json.loads(l)['train_step']['ms_per_step'],4)) for l in sys.stdin if l.startswith('{')]"
  done
done
