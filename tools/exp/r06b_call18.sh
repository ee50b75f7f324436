for bp in 4 3 2; do python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-pmc --opt lbs.blocks_per_cu=$bp 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())['digest']; print('bp=$bp', {k:d[k] for k in d if k.startswith('scene') and ('frame_ms' in k or 'skin_ms' in k)}, 'c3', d['c3_frame_ms'])"; done
