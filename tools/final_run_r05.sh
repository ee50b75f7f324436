#!/bin/bash
# Round 5, the final library on one box: the GPU suite, smoke, the line with default arguments and with the driver's, the 256-character
# scene's kernels under rocprofv3.  Everything lands in gpurun_out/final/ (copied to profiles/r05_final_* by hand).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/final; mkdir -p $O
cd $R
if [ "$1" != "--no-tests" ]; then
  timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; grep -i "smoke" $O/smoke.txt | tail -1
else
  timeout 600 python -m pytest tests/test_frame_skin_gpu.py tests/test_anim_gpu.py -q -m gpu -k "scene or pipelined or steady" > $O/pytest_scene.txt 2>&1; grep -E "passed|failed|error" $O/pytest_scene.txt | tail -3
fi
timeout 600 python bench.py > $O/bench_plain.json 2> $O/bench_plain.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scene -o scene -- python $R/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > $O/scene_under_trace.json 2> $O/scene.err )
find $O/scene -name "*kernel_stats.csv" -exec cp {} $O/scene_kernel_stats.csv \;
rm -rf $O/scene
head -8 $O/scene_kernel_stats.csv
python - <<PY
import json
for f in ("bench_plain","bench_driver_args"):
    d=json.load(open("$O/%s.json"%f))
    e=d.get("extra",{})
    print(f, d["value"], d["roofline"]["frac"], d["roofline"].get("frac_at_6_sets"), {k:(e.get(k) or {}).get("frame_ms") for k in ("c2","c3","c5","scene_256x1","scene_64x4")})
PY
