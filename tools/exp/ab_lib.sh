#!/bin/bash
# GPU box: the same command with the product library and with another build of it (FYX_LIB_PATH), alternating, on ONE box --
# box-to-box differences (+-8 % on the scene ticks) are larger than most of what an A/B is asked about.
#   tools/exp/ab_lib.sh tools/exp/libs/libfyrox_hip_prev.so 3 python tools/bench_scene.py --characters 256 --instances 1 --verts 5000
OTHER=$1; N=$2; shift 2
for i in $(seq $N); do
  echo "product: $("$@" 2>/dev/null | head -c ${AB_CHARS:-420})"
  echo "other:   $(FYX_LIB_PATH=$(pwd)/$OTHER "$@" 2>/dev/null | head -c ${AB_CHARS:-420})"
done
