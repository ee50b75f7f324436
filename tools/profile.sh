#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the bench command + separate PMC passes.
# Outputs under gpurun_out/prof/ ; summaries are copied into profiles/ by tools/summarize_profile.py.
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
BENCH="python $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -o bench -- $BENCH > "$ROOT/$OUT/bench_under_trace.json" 2> "$ROOT/$OUT/trace.err" )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_s1" -o bench -- $BENCH --opt lbs.streams=1 > "$ROOT/$OUT/bench_under_trace_s1.json" 2> "$ROOT/$OUT/trace_s1.err" )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_fetch" -o pmc -- python $ROOT/tools/pmc_probe.py > /dev/null 2> "$ROOT/$OUT/pmc_fetch.err" )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_write" -o pmc -- python $ROOT/tools/pmc_probe.py > /dev/null 2> "$ROOT/$OUT/pmc_write.err" )
( cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_lds" -o pmc -- python $ROOT/tools/pmc_probe.py > /dev/null 2> "$ROOT/$OUT/pmc_lds.err" )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_scene_fetch" -o pmc -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 20 --batched-only > /dev/null 2> "$ROOT/$OUT/pmc_scene_fetch.err" )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_scene_write" -o pmc -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 20 --batched-only > /dev/null 2> "$ROOT/$OUT/pmc_scene_write.err" )
python $ROOT/bench.py --steps 2000 --warmup 100 --cpu-seconds 3 > "$ROOT/$OUT/bench_plain.json" 2> "$ROOT/$OUT/bench_plain.err"
python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$ROOT/$OUT/bench_driver_args.json" 2> "$ROOT/$OUT/bench_driver_args.err"
# C3 crowd, pose path + instanced skinning (kernel-trace only)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_pose" -o pose -- python $ROOT/tools/bench_pose.py --frames 200 > "$ROOT/$OUT/pose_under_trace.json" 2> "$ROOT/$OUT/trace_pose.err" )
# the crowd skinning launch ALONE (no pose kernels, no uploads beside it): exact then fused, 10 + 3 x 120 launches each
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_crowd_lone" -o crowd -- python $ROOT/tools/exp/crowd_time.py > "$ROOT/$OUT/crowd_lone_under_trace.jsonl" 2> "$ROOT/$OUT/trace_crowd_lone.err" )
python $ROOT/tools/exp/crowd_time.py > "$ROOT/$OUT/crowd_lone.jsonl" 2> "$ROOT/$OUT/crowd_lone.err"
python $ROOT/tools/bench_pose.py > "$ROOT/$OUT/pose_plain.json" 2> "$ROOT/$OUT/pose_plain.err"
# single characters (C2, C5): kernel trace + the frame's timeline (kernel durations and the gaps between them), and the time stamps
# inside the update kernel (a build of the library with stamps, tools/exp/upd_stamps_build.sh, if it travelled with the snapshot)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_character" -o chr -- python $ROOT/tools/bench_character.py > "$ROOT/$OUT/character_under_trace.json" 2> "$ROOT/$OUT/trace_character.err" )
F=$(find "$OUT/trace_character" -name "*_kernel_trace.csv" | head -1)
[ -n "$F" ] && python $ROOT/tools/frame_timeline.py "$F" > "$ROOT/$OUT/character_timeline.json" 2> "$ROOT/$OUT/character_timeline.err"
python $ROOT/tools/bench_character.py > "$ROOT/$OUT/character_plain.json" 2> "$ROOT/$OUT/character_plain.err"
[ -f $ROOT/tools/exp/libs/libfyrox_hip_updstamp.so ] && FYX_LIB_PATH=$ROOT/tools/exp/libs/libfyrox_hip_updstamp.so python $ROOT/tools/exp/upd_stamps.py > "$ROOT/$OUT/update_stamps.json" 2> "$ROOT/$OUT/update_stamps.err"
python $ROOT/tools/bench_pose.py --palette-output > "$ROOT/$OUT/pose_palette_output.json" 2> "$ROOT/$OUT/pose_palette_output.err"
python $ROOT/tools/bench_pose.py --root-motion > "$ROOT/$OUT/pose_root_motion.json" 2> "$ROOT/$OUT/pose_root_motion.err"
# extended launches: blend shapes, vertex-buffer-in / vertex-buffer-out
python $ROOT/tools/bench_ex.py > "$ROOT/$OUT/bench_ex.json" 2> "$ROOT/$OUT/bench_ex.err"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_ex" -o ex -- python $ROOT/tools/bench_ex.py --streams 1 --steps 300 > "$ROOT/$OUT/bench_ex_under_trace.json" 2> "$ROOT/$OUT/trace_ex.err" )
# heterogeneous scenes: many animators / meshes per frame, batched vs one by one
python $ROOT/tools/bench_scene.py > "$ROOT/$OUT/scene_64x4.json" 2> "$ROOT/$OUT/scene_64x4.err"
python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 > "$ROOT/$OUT/scene_256x1.json" 2> "$ROOT/$OUT/scene_256x1.err"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_scene" -o scene -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > "$ROOT/$OUT/scene_under_trace.json" 2> "$ROOT/$OUT/trace_scene.err" )
python $ROOT/tools/bench_pose.py --opt lbs.streams=1 --opt lbs.exact=0 > "$ROOT/$OUT/pose_fused.json" 2> "$ROOT/$OUT/pose_fused.err"
python $ROOT/tools/write_ceiling.py > "$ROOT/$OUT/write_ceiling.json" 2> "$ROOT/$OUT/write_ceiling.err"
python $ROOT/tools/calib.py --rounds 2 > "$ROOT/$OUT/calibration_stream.json" 2> "$ROOT/$OUT/calib.err"
find "$OUT" -name "*_kernel_trace.csv" -size +12M -delete   # raw per-dispatch traces of the long runs do not travel back
find "$OUT" -name "*.csv" | head -40
du -sh "$OUT"
