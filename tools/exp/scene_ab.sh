#!/bin/bash
# GPU box: kernel-trace averages of the 256-character scene's kernels for the product library and for variant libraries.
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/sab
for tag in product "$@" product; do
  if [ "$tag" = product ]; then unset FYX_LIB_PATH; else export FYX_LIB_PATH=$ROOT/tools/exp/libs/libfyrox_hip_$tag.so; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/sab/$tag" -o s -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 60 --batched-only > "$ROOT/gpurun_out/sab/$tag.json" 2> /dev/null )
  echo "== $tag"; grep -E "scene_kernel|lbs_skin_batch" gpurun_out/sab/$tag/s_kernel_stats.csv | python3 -c "import csv,sys; [print('  ', r[0][:64], r[1], round(float(r[3])/1000,2)) for r in csv.reader(sys.stdin)]"
  python3 -c "import json; d=json.loads(open('gpurun_out/sab/$tag.json').read().strip().splitlines()[-1]); print('   frame %.4f pose %.4f skin %.4f' % (d['frame_ms_wall'], d['pose_ms_wall'], d['skin_ms_wall']))"
  find gpurun_out/sab -name "*kernel_trace.csv" -delete
done
