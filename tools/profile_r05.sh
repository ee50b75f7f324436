#!/bin/bash
# Round 5 profile run (GPU box, through gpurun): what profiles/r05_* is made of.  tools/summarize_r05.py condenses gpurun_out/prof05.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/prof05
mkdir -p $OUT
HEAD="python $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras --no-pmc"
# 1. the headline leg three times, each process under the kernel trace AND reading its own per-dispatch events (one launch stream)
for k in 1 2; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/head$k" -o bench -- $HEAD --opt lbs.streams=1 > "$ROOT/$OUT/head$k.json" 2> "$ROOT/$OUT/head$k.err" )
done
# ... and once untraced with the same arguments (events only), once on the library's two launch streams under the trace
$HEAD --opt lbs.streams=1 > $OUT/head_untraced.json 2> $OUT/head_untraced.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/streams2" -o bench -- $HEAD > "$ROOT/$OUT/streams2.json" 2> "$ROOT/$OUT/streams2.err" )
# 2. the crowd launch alone (exact, fused), traced and untraced; the C3 frame's kernels (one chain on one stream)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/crowd_lone" -o crowd -- python $ROOT/tools/exp/crowd_time.py > "$ROOT/$OUT/crowd_lone_under_trace.jsonl" 2> "$ROOT/$OUT/crowd_lone.err" )
python tools/exp/crowd_time.py > $OUT/crowd_lone.jsonl 2>> $OUT/crowd_lone.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/pose" -o pose -- python $ROOT/tools/bench_pose.py --frames 200 --palette-output > "$ROOT/$OUT/pose_under_trace.json" 2> "$ROOT/$OUT/pose.err" )
python tools/exp/r04_timeline.py > $OUT/c3_timeline.jsonl 2> $OUT/c3_timeline.err
# 2b. one character's frame with the skinning inside the pose launch: the forms side by side, and the launch under the trace
python tools/exp/r05_frame_skin.py 400 > $OUT/frame_skin.jsonl 2> $OUT/frame_skin.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/frame_skin" -o fs -- python $ROOT/tools/exp/r05_frame_skin.py 200 > "$ROOT/$OUT/frame_skin_under_trace.jsonl" 2>> "$ROOT/$OUT/frame_skin.err" )
python tools/exp/r05_scene.py > $OUT/scene_records.jsonl 2> $OUT/scene_records.err
python tools/exp/r05_vertex_buffer.py > $OUT/vertex_buffer.jsonl 2> $OUT/vertex_buffer.err
# 3. counters, one pass per group: HBM bytes and LDS conflicts, coherent and random bone indices (C4, C3)
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $G | cut -d' ' -f1)
  ( cd /tmp && rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_$tag" -o pmc -- python $ROOT/tools/pmc_probe_r04.py > /dev/null 2> "$ROOT/$OUT/pmc_$tag.err" )
done
# 4. single characters and the 256-character scene: kernel durations
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/character" -o chr -- python $ROOT/tools/bench_character.py > "$ROOT/$OUT/character_under_trace.json" 2> "$ROOT/$OUT/character.err" )
python tools/bench_character.py > $OUT/character_plain.json 2>> $OUT/character.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/scene" -o scene -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > "$ROOT/$OUT/scene_under_trace.json" 2> "$ROOT/$OUT/scene.err" )
# 5. the lines: default arguments, the driver's arguments, and two ranks on this one GPU (test hook: every N > 1 key)
python bench.py --cpu-seconds 3 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
FYX_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 20 --warmup 5 --sets 4 > $OUT/bench_2ranks_one_gpu_test_hook.json 2> $OUT/bench_2ranks.err
FYX_BENCH_DEVICE=0 python bench.py --gpus 2 --one-process --steps 20 --warmup 5 --sets 2 > $OUT/bench_one_process_one_gpu_test_hook.json 2> $OUT/bench_one_process.err
find "$OUT" -name "*_kernel_trace.csv" -size +6M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"; ls $OUT | head -50
