set -u
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
python $ROOT/tools/bench_scene.py > "$ROOT/$OUT/scene_64x4.json" 2> "$ROOT/$OUT/scene_64x4.err"
python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 > "$ROOT/$OUT/scene_256x1.json" 2> "$ROOT/$OUT/scene_256x1.err"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace_scene" -o scene -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > "$ROOT/$OUT/scene_under_trace.json" 2> "$ROOT/$OUT/trace_scene.err" )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_scene_fetch" -o pmc -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 20 --batched-only > /dev/null 2> "$ROOT/$OUT/pmc_scene_fetch.err" )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_scene_write" -o pmc -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 20 --batched-only > /dev/null 2> "$ROOT/$OUT/pmc_scene_write.err" )
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
grep scene_kernel $OUT/trace_scene/scene_kernel_stats.csv | cut -c1-120
