#!/bin/bash
# Host-side AddressSanitizer run WITH the GPU (stream / event bookkeeping of fyx_api.hip, the control-block double
# buffering, the batch tables, the comm layer's host code): the library's host code is instrumented, device code is not
# (-fno-gpu-sanitize).  Two steps, because the GPU box is not the build box:
#   tools/asan_gpu.sh build          (here)   -> tools/exp/libs/libfyrox_hip_asan.so
#   tools/asan_gpu.sh run                     (GPU box, through gpurun) -> gpurun_out/asan_gpu_cpp_host.log
# The driver is the compiled C++ host (tests/cpp/host_parity.cpp: C1, a clip, then a 16-frame engine-style loop --
# scene update with alternating palette outputs, worker streams, batches with a changing job table, pipelined and
# joined frames, per-launch timing), itself built with ASan.  The Python test-suite cannot be the driver: an
# un-instrumented python with a preloaded ASan runtime dies inside the HIP runtime's start-up (hipInit segfaults before
# any library code runs; measured with and without torch's bundled runtime), so `pytest -m gpu` under ASan is not possible here.
# protect_shadow_gap=0: the ROCm runtime maps device-visible memory where ASan would like to keep its shadow gap;
# allocator_may_return_null=1: the runtime probes with huge allocations and copes with a null.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/tools/exp/libs/libfyrox_hip_asan.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
case "${1:-build}" in
build)
  OUT=/tmp/fyx_asan_gpu; mkdir -p $OUT $ROOT/tools/exp/libs
  for f in anim_api fyx_api comm_api anim_kernels lbs_kernels; do
    $HIPCC -O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -Wno-unused-function \
           -I$ROOT/include -c $ROOT/fyrox_amd/csrc/$f.hip -o $OUT/$f.o &
  done; wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -o $LIB $OUT/*.o && ls -la $LIB
  # the compiled C++ host (no Python, no torch) against the instrumented library
  make -s -C $ROOT/oracle
  cp $LIB $OUT/libfyrox_hip.so
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address -shared-libasan $ROOT/tests/cpp/host_parity.cpp \
      -o $ROOT/tools/exp/libs/host_parity_asan -L$OUT -L$ROOT/oracle -lfyrox_hip -lfyrox_oracle \
      -Wl,-rpath,'$ORIGIN/asan_lib' -Wl,-rpath,$ROOT/oracle -Wl,-rpath,/opt/rocm/lib && mkdir -p $ROOT/tools/exp/libs/asan_lib && cp $LIB $ROOT/tools/exp/libs/asan_lib/libfyrox_hip.so ;;
run)
  RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  mkdir -p $ROOT/gpurun_out
  ( cd $ROOT && LD_LIBRARY_PATH=$(dirname $RT):$ROOT/oracle ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0:allocator_may_return_null=1 \
      timeout 300 tools/exp/libs/host_parity_asan > gpurun_out/asan_gpu_cpp_host.log 2>&1; echo "rc=$?" >> gpurun_out/asan_gpu_cpp_host.log; tail -5 gpurun_out/asan_gpu_cpp_host.log )
  ;;
esac
