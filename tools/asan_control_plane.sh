#!/bin/bash
# Host-side memory check of the control plane (no GPU needed): builds an AddressSanitizer variant of libfyrox_hip.so
# into /tmp (device code is not instrumented), swaps it in for the run and puts the real library back afterwards,
# then runs the control-only tests -- planners (threaded crowd / scene planning included), event queues, builders,
# 40 fuzzed machines, run-time edits of machines (clear + rebuild + state restore) -- under it.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/fyx_asan; mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for f in anim_api fyx_api comm_api anim_kernels lbs_kernels host_geom; do
  $HIPCC -O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -Wno-unused-function \
         -c $ROOT/fyrox_amd/csrc/$f.hip -o $OUT/$f.o &
done; wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -o $OUT/libfyrox_hip.so $OUT/*.o || exit 1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cp $ROOT/fyrox_amd/libfyrox_hip.so $OUT/libfyrox_hip.real.so
trap 'cp $OUT/libfyrox_hip.real.so $ROOT/fyrox_amd/libfyrox_hip.so' EXIT
cp $OUT/libfyrox_hip.so $ROOT/fyrox_amd/libfyrox_hip.so
# (a command of the caller's instead of the tests: `tools/asan_control_plane.sh python tools/fuzz_api_host.py --count 300`)
if [ $# -gt 0 ]; then cd $ROOT && LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 "$@"; exit $?; fi
cd $ROOT && LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest tests/test_anim_control.py tests/test_machine_edits.py tests/test_abi.py -x -q -p no:cacheprovider
