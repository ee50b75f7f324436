for bp in 4 6 8 12 16 4 8; do python tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 100 --batched-only --opt lbs.blocks_per_cu=$bp 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('5k bp=$bp', {k:round(d[k],5) for k in ('frame_ms_gpu','pose_ms_gpu','skin_ms_gpu')})"; done
for bp in 4 8 16; do python tools/bench_scene.py --characters 64 --instances 4 --verts 20000 --frames 60 --batched-only --opt lbs.blocks_per_cu=$bp 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('64x4 bp=$bp', {k:round(d[k],5) for k in ('frame_ms_gpu','pose_ms_gpu','skin_ms_gpu')})"; done
for bp in 4 8 16; do python tools/bench_scene.py --characters 256 --instances 1 --verts 20000 --frames 60 --batched-only --opt lbs.blocks_per_cu=$bp 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('256x20k bp=$bp', {k:round(d[k],5) for k in ('frame_ms_gpu','pose_ms_gpu','skin_ms_gpu')})"; done
