O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_lbs_gpu.py -q -x -k "batch" 2>&1 | tail -3
for dyn in 0 1 0 1; do python tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 100 --batched-only --opt lbs.dyn=$dyn 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('dyn=$dyn', {k:round(d[k],5) for k in ('frame_ms_gpu','pose_ms_gpu','skin_ms_gpu','skin_algorithmic_GBps')})"; done
for dyn in 0 1; do python tools/bench_scene.py --characters 64 --instances 4 --verts 20000 --frames 60 --batched-only --opt lbs.dyn=$dyn 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('64x4 dyn=$dyn', {k:round(d[k],5) for k in ('frame_ms_gpu','pose_ms_gpu','skin_ms_gpu','skin_algorithmic_GBps')})"; done
for dyn in 0 1; do python tools/bench_scene.py --characters 256 --instances 1 --verts 20000 --frames 60 --batched-only --opt lbs.dyn=$dyn 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('256x1x20k dyn=$dyn', {k:round(d[k],5) for k in ('frame_ms_gpu','pose_ms_gpu','skin_ms_gpu','skin_algorithmic_GBps')})"; done
