mkdir -p gpurun_out/r06b
python tools/bench_ex.py 2>/dev/null | tail -1 > gpurun_out/r06b/bench_ex.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06b/bench_ex.json"))
for k,v in d["results"].items():
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()} if isinstance(v,dict) else v)
PY
