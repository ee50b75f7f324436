#!/bin/bash
# GPU box: kernel-trace averages of the C3 crowd's pose kernels under an option's values.  Usage: pose_ab.sh OPTION V1 V2 ...
opt=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/ab
for v in "$@"; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/ab/$opt$v" -o p -- python $ROOT/tools/bench_pose.py --instances ${INSTANCES:-1000} --frames 150 --opt lbs.streams=1 --opt $opt=$v > "$ROOT/gpurun_out/ab/$opt$v.json" 2> /dev/null )
  echo "== $opt=$v"; grep -E "pose_|palette" gpurun_out/ab/$opt$v/p_kernel_stats.csv | python3 -c "import csv,sys; [print(r[0][:70], r[1], round(float(r[3])/1000,2)) for r in csv.reader(sys.stdin)]"
  find gpurun_out/ab -name "*kernel_trace.csv" -delete
done
