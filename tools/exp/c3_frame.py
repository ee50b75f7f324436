import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd, bench
ctx = fyrox_amd.Context(0)
ex = bench.extras(ctx)
for k, r in ex.items():
    print(k, {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in r.items() if kk.endswith("_ms") or kk.startswith("frame_ms")}, r["parity"]["bit_exact"], r["parity"]["end_to_end_max_rel_err"], flush=True)
