#!/bin/bash
# Round 6 profile run (GPU box, through gpurun): what profiles/r06_* (outside the study directories) is made of.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof06
mkdir -p $OUT
HEAD="python $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras --no-pmc"
# 1. the headline's kernel under the kernel trace: the lone launch (one launch stream) and the bench's own two streams; and the driver's
#    command line itself under the trace (every kernel of the run)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/streams1 -o bench -- $HEAD --opt lbs.streams=1 > $OUT/streams1.json 2> $OUT/streams1.err )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/streams2 -o bench -- $HEAD > $OUT/streams2.json 2> $OUT/streams2.err )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver -o bench -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --full-record $OUT/bench_driver_args_under_trace_full.json > $OUT/bench_driver_args_under_trace.json 2> $OUT/driver.err )
# 2. the 256-character scene's kernels, one character's frames
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scene -o scene -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > $OUT/scene_under_trace.json 2> $OUT/scene.err )
python tools/bench_character.py > $OUT/character_plain.json 2> $OUT/character.err
# 3. two ranks on this one GPU (test hook: every N > 1 key), both roads
FYX_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 20 --warmup 5 --sets 4 --full-record $OUT/bench_2ranks_full.json > $OUT/bench_2ranks_one_gpu_test_hook.json 2> $OUT/bench_2ranks.err
FYX_BENCH_DEVICE=0 python bench.py --gpus 2 --one-process --steps 20 --warmup 5 --sets 2 --full-record $OUT/bench_one_process_full.json > $OUT/bench_one_process_one_gpu_test_hook.json 2> $OUT/bench_one_process.err
for d in streams1 streams2 driver scene; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; rm -rf $OUT/$d; done
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
ls -la $OUT; head -4 $OUT/streams1_kernel_stats.csv | cut -c1-220; head -6 $OUT/scene_kernel_stats.csv | cut -c1-220; cat $OUT/bench_2ranks_one_gpu_test_hook.json | cut -c1-1800
