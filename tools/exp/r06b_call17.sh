for dyn in 0 1 0 1; do python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-pmc --opt lbs.dyn=$dyn 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())['digest']; print('dyn=$dyn', {k:d[k] for k in d if k.startswith('scene') and ('frame_ms' in k or 'skin_ms' in k)}, 'c3', d['c3_frame_ms'])"; done
