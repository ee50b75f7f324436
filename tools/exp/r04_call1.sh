#!/bin/bash
# Round 4, first GPU call: regression check of the refactored pose path, the C3 frame under the new stream / upload options, the two
# untried crowd-kernel forms, a kernel trace of the pipelined frame, the worst-case bone indices.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/r04c1
mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
timeout 400 python tools/exp/r04_frame.py > $OUT/frame.jsonl 2> $OUT/frame.err; echo "frame rc $?"; cat $OUT/frame.jsonl; tail -3 $OUT/frame.err
timeout 300 python tools/exp/r04_crowd_forms.py > $OUT/crowd_forms.jsonl 2> $OUT/crowd_forms.err; echo "crowd rc $?"; cat $OUT/crowd_forms.jsonl; tail -3 $OUT/crowd_forms.err
for lean in 0 1; do
( cd /tmp && WHICH=trace FRAMES=100 CROWD_LEAN=$lean timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace$lean -o pipe -- python $ROOT/tools/exp/r04_frame.py > $ROOT/$OUT/trace$lean.jsonl 2> $ROOT/$OUT/trace$lean.err )
F=$(find $OUT/trace$lean -name "*kernel_trace.csv" | head -1)
[ -n "$F" ] && python tools/exp/r04_overlap.py $F 600 | tee $OUT/overlap$lean.json
cat $OUT/trace$lean.jsonl
done
timeout 200 python bench.py --random-bones --no-extras --no-cpu-baseline --steps 500 > $OUT/bench_random.json 2> $OUT/bench_random.err; echo "bench rc $?"
python - <<'P'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r04c1/bench_random.json") if l.startswith("{")][-1])
    print("random bones:", d["value"], d["roofline"]["kernel_us"], d["roofline"]["frac"], d["parity"])
except Exception as e:
    print("bench_random:", e)
P
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
