#!/bin/bash
# Build tools/exp/libs/libfyrox_hip_r06sstamp.so: the product sources compiled with -DFYX_SCENE_STAMPS (wall-clock stamps of every workgroup
# of the scene's sampler, read by tools/exp/r06_scene_stamps.py through fyx_exp_scene_stamps).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf /tmp/var6 && mkdir -p /tmp/var6/fyrox_amd && cp -r "$ROOT/fyrox_amd/csrc" /tmp/var6/fyrox_amd/ && cp -r "$ROOT/include" /tmp/var6/
cd /tmp/var6/fyrox_amd/csrc && rm -rf build
make -j8 FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-cuda-compat -DFYX_SCENE_STAMPS $EXTRA" 2>&1 | grep -E "error" && exit 1
mkdir -p "$ROOT/tools/exp/libs" && cp ../libfyrox_hip.so "$ROOT/tools/exp/libs/libfyrox_hip_r06sstamp.so"
echo built
