#!/bin/bash
# Round 4, second profile run (GPU box, through gpurun): the legs whose kernels changed after tools/profile_r04.sh was run -- one
# character's frame in one launch with the wide walk, the scene tick, the crowd's packed update -- and the lines.  Summaries are
# copied into profiles/r04b_* by hand (kernel stats CSVs as they are).
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/prof04b
mkdir -p $OUT
# 1. single characters and the 256-character scene: kernel durations, and the same untraced
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/character" -o chr -- python $ROOT/tools/bench_character.py > "$ROOT/$OUT/character_under_trace.json" 2> "$ROOT/$OUT/character.err" )
timeout 200 python tools/bench_character.py > $OUT/character_plain.json 2>> $OUT/character.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/scene" -o scene -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > "$ROOT/$OUT/scene_under_trace.json" 2> "$ROOT/$OUT/scene.err" )
timeout 200 python tools/exp/r04_scene.py > $OUT/scene_plain.json 2>> $OUT/scene.err
# 2. the C3 frame's kernels (one chain on one stream) and its timeline with alternating frame streams
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/pose" -o pose -- python $ROOT/tools/bench_pose.py --frames 200 --palette-output > "$ROOT/$OUT/pose_under_trace.json" 2> "$ROOT/$OUT/pose.err" )
timeout 200 python tools/exp/r04_timeline.py > $OUT/c3_timeline.jsonl 2> $OUT/c3_timeline.err
# 3. the lines: default arguments and the driver's arguments
timeout 500 python bench.py --cpu-seconds 3 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
find "$OUT" -name "*_kernel_trace.csv" -size +6M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"; ls $OUT | head -50
