#!/bin/bash
# GPU box: the PMC passes and the C3 pose trace of tools/profile.sh alone (into gpurun_out/prof, next to an earlier full run)
set -u
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_fetch" -o pmc -- python $ROOT/tools/pmc_probe.py > /dev/null 2> "$ROOT/$OUT/pmc_fetch.err" )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_write" -o pmc -- python $ROOT/tools/pmc_probe.py > /dev/null 2> "$ROOT/$OUT/pmc_write.err" )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_pose" -o pose -- python $ROOT/tools/bench_pose.py --frames 200 > "$ROOT/$OUT/pose_under_trace.json" 2> "$ROOT/$OUT/trace_pose.err" )
python $ROOT/tools/bench_pose.py > "$ROOT/$OUT/pose_plain.json" 2> "$ROOT/$OUT/pose_plain.err"
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
head -4 $OUT/trace_pose/pose_kernel_stats.csv | cut -c1-130
