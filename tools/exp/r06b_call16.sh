O=gpurun_out/r06b; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gputests3.txt 2>&1; grep -n "passed\|failed" $O/gputests3.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python tools/bench_ex.py 2>/dev/null | tail -1 > $O/bench_ex.json; python -c "
import json; d=json.load(open('$O/bench_ex.json'))['results']
print({k: round(v['us_per_launch'],2) for k,v in d.items() if isinstance(v,dict) and 'us_per_launch' in v})"
