#!/bin/bash
# Experiment build of the library with extra compiler flags: scripts/build_variant.sh <name> [-DFOO=1 ...] -> localrf_amd/csrc/liblrf_<name>.so
# (select it with LRF_LIB=<path>; scripts/ab_shade.sh interleaves two builds on one box)
NAME=$1; shift
cd "$(dirname "$0")/../localrf_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I ../../include "$@" -shared -o liblrf_$NAME.so lrf_render.hip && echo built liblrf_$NAME.so
