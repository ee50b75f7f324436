for st in 2 3 4 2 3 4; do python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-extras --no-pmc --opt lbs.streams=$st 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('streams=$st value', d['value'], 'ms', d['ms_per_step'], 'overlapped', d['roofline'].get('overlapped'))"; done
