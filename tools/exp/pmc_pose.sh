#!/bin/bash
# GPU box: HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) of the C3 crowd frame's kernels.
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/pmcp
for cn in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $cn --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmcp/$cn" -o pmc -- python $ROOT/tools/bench_pose.py --frames 30 --palette-output "$@" > /dev/null 2>&1 )
  python3 - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$ROOT/gpurun_out/pmcp/$cn/pmc_counter_collection.csv")):
    if r["Counter_Name"] == "$cn":
        agg[r["Kernel_Name"].split("(")[0].split("::")[-1][:40]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    v.sort(); print("$cn", k, "launches", len(v), "median KB", round(v[len(v)//2], 1))
PY
done
find gpurun_out/pmcp -name "*kernel_trace.csv" -delete
