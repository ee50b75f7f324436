#!/bin/bash
# GPU box: dynamic instruction counts (SQ) of the C3 frame's kernels, serial frames -- what each kernel ISSUES, the currency the
# kernels of a pipelined frame compete in.  One rocprofv3 pass per counter group; medians per kernel.
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/r04_pmc_pose
mkdir -p $OUT
G1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES"
G2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -o pmc -- python $ROOT/tools/bench_pose.py --frames 12 --warmup 4 --palette-output --opt lbs.streams=1 > "$ROOT/$OUT/g$i.log" 2>&1 )
  python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$ROOT/$OUT/g$i/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1][:44]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if any(s in k for s in ("pose_", "lbs_skin", "ctrl_copy")):
        print(k, {c: sorted(v)[len(v) // 2] for c, v in sorted(d.items())}, "launches", len(next(iter(d.values()))))
PY
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
