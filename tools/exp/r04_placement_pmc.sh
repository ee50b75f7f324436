#!/bin/bash
# GPU box: what differs between a FAST and a SLOW placement of the crowd kernel's three output streams (fused mode, 60 vs 78 us)?
# Address-translation and memory-side counters per dispatch, eight separate allocations held at once (WHICH=pmc in r04_fused_placement.py:
# three launches per allocation, in allocation order), one rocprofv3 pass per counter group; and the same launches timed without counters.
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/r04_placement_pmc
mkdir -p $OUT
( cd /tmp && rocprofv3 -L > "$ROOT/$OUT/counters.txt" 2>&1 )
grep -oE "(TCP_UTCL1|TCC_EA0|TCC_[A-Z_]*STALL|TCP_TCC|UTCL2|TCC_TAG|TCC_WRITE|TCC_MC|GRBM_GUI|TCP_PENDING)[A-Za-z0-9_]*" $OUT/counters.txt | sort -u | head -80 > $OUT/candidates.txt
WHICH=pmc python tools/exp/r04_fused_placement.py > $OUT/timing.jsonl 2> $OUT/timing.err
i=0
for G in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_PERMISSION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && WHICH=pmc rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$ROOT/$OUT/g$i" -o pmc -- python $ROOT/tools/exp/r04_fused_placement.py > "$ROOT/$OUT/g$i.log" 2>&1 )
  python3 - <<PY
import csv, collections, glob
rows = collections.defaultdict(dict)
for f in glob.glob("$ROOT/$OUT/g$i/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lbs_skin_crowd" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
print("group $i:", "$G")
# three launches per allocation, allocations in order: report the LAST launch of each allocation
for k in range(0, min(8, len(ids) // 3)):
    d = rows[ids[3 * k + 2]]
    print("  allocation", k, {c: v for c, v in sorted(d.items())})
PY
done
cat $OUT/timing.jsonl
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
