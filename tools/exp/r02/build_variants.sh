#!/bin/bash
# Kernel experiments: rebuild lbs_kernels.hip with -D switches and link each against the product's other objects into
# tools/exp/libs/libfyrox_hip_<tag>.so (load with FYX_LIB_PATH=...).  Usage: build_variants.sh TAG "-DFOO=1" [TAG "-D..."]...
set -e
cd "$(dirname "$0")/../../fyrox_amd/csrc"
make -s
mkdir -p ../../tools/exp/libs build/exp
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-cuda-compat $defs -c lbs_kernels.hip -o build/exp/lbs_kernels_$tag.o &
done
wait
for o in build/exp/lbs_kernels_*.o; do
  tag=$(basename $o .o); tag=${tag#lbs_kernels_}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libs/libfyrox_hip_$tag.so $o $(ls build/*.o | grep -v lbs_kernels.o)
done
ls -la ../../tools/exp/libs
