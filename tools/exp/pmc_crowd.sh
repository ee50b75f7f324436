#!/bin/bash
# GPU box: SQ / TCP counters of the crowd skinning kernel on C3, exact and fused, one rocprofv3 pass per counter group.
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/pmc_crowd
mkdir -p $OUT
( cd /tmp && rocprofv3 -L > "$ROOT/$OUT/counters.txt" 2>&1 )
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
G2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES"
G3="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr"
G4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  for EX in 1 0; do
    ( cd /tmp && REPS=10 OPTS="lbs.exact=$EX $EXTRA_OPTS" rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$ROOT/$OUT/g${i}_e$EX" -o pmc -- python $ROOT/tools/exp/crowd_pmc_run.py > "$ROOT/$OUT/g${i}_e$EX.log" 2>&1 )
    python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(list)
for f in glob.glob("$ROOT/$OUT/g${i}_e$EX/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "crowd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v.sort(); print("exact=$EX", k, "launches", len(v), "median", v[len(v)//2])
PY
  done
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
