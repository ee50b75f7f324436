#!/bin/bash
# Build tools/exp/libs/libfyrox_hip_r05stamp.so: the product sources compiled with -DFYX_FRAME_STAMPS (wall-clock stamps of every workgroup
# of the one-launch frame, read by tools/exp/r05_stamps.py through fyx_exp_frame_stamps).  Extra flags: $EXTRA.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf /tmp/var5 && mkdir -p /tmp/var5/fyrox_amd && cp -r "$ROOT/fyrox_amd/csrc" /tmp/var5/fyrox_amd/ && cp -r "$ROOT/include" /tmp/var5/
cd /tmp/var5/fyrox_amd/csrc && rm -rf build
make -j8 FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-cuda-compat -DFYX_FRAME_STAMPS $EXTRA" 2>&1 | grep -E "error" && exit 1
mkdir -p "$ROOT/tools/exp/libs" && cp ../libfyrox_hip.so "$ROOT/tools/exp/libs/libfyrox_hip_r05stamp.so"
echo built
