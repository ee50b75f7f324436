#!/bin/bash
# GPU box: pose_update_kernel's average duration for one character (one wave) under ablation variants of the library.
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/abl
for tag in product "$@" product; do
  if [ "$tag" = product ]; then unset FYX_LIB_PATH; else export FYX_LIB_PATH=$ROOT/tools/exp/libs/libfyrox_hip_$tag.so; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/abl/$tag" -o p -- python $ROOT/tools/bench_pose.py --instances ${INSTANCES:-1} --frames 150 --palette-output > /dev/null 2>&1 )
  echo "== $tag: $(grep -E "pose_update" gpurun_out/abl/$tag/p_kernel_stats.csv | python3 -c "import csv,sys; [print(r[1], round(float(r[3])/1000,2), 'us') for r in csv.reader(sys.stdin)]")"
  find gpurun_out/abl -name "*kernel_trace.csv" -delete
done
