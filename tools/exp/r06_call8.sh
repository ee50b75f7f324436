#!/bin/bash
# Round 6: the cursor sampler -- correctness first (the pose-path GPU tests), then the scene's kernel times and the one-character frames.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_cursor; mkdir -p $O
timeout 1500 python -m pytest tests/test_anim_gpu.py tests/test_frame_skin_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt | cut -c1-300
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scene -o scene -- python $R/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > $O/scene_under_trace.json 2> $O/scene.err )
f=$(find $O/scene -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/scene_kernel_stats.csv && head -12 $O/scene_kernel_stats.csv | cut -c1-200
timeout 600 python tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 150 --batched-only > $O/scene_plain.json 2>> $O/scene.err; cat $O/scene_plain.json | cut -c1-1500
timeout 600 python tools/bench_character.py > $O/character.json 2> $O/character.err; cat $O/character.json | cut -c1-1500
