set -u
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/ranked
SETS="anim.sample_ranked=1;anim.sample_ranked=0" timeout 300 python tools/exp/r04_frame_ab.py > gpurun_out/ranked/frame.jsonl 2>&1
cat gpurun_out/ranked/frame.jsonl
for r in 1 0; do
  for C in WRITE_SIZE FETCH_SIZE; do
    ( cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pm_${r}_$C -o pmc -- python $ROOT/tools/bench_pose.py --frames 12 --warmup 4 --palette-output --opt lbs.streams=1 --opt anim.sample_ranked=$r > /dev/null 2> /tmp/pm_${r}_$C.err )
    f=$(find /tmp/pm_${r}_$C -name "*counter_collection.csv" | head -n 1)
    if [ -n "$f" ]; then python - "$f" $r $C <<'PY'
import csv, sys, collections
f, r, C = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if row.get("Counter_Name") == C:
        acc[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    if "pose_" in k or "crowd" in k:
        v = sorted(v); print(f"ranked={r} {C} {k}: median {v[len(v)//2]:.0f} over {len(v)} dispatches")
PY
    fi
  done
done
