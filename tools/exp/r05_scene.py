#!/usr/bin/env python3
"""bench.py's scene records alone (256 x 1 and 64 x 4), options from the command line (key=value)."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("fyx_bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
argv, sys.argv = sys.argv, ["bench.py"]
spec.loader.exec_module(b)
sys.argv = argv
import fyrox_amd
with fyrox_amd.Context(0) as ctx:
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    for n_chars, n_inst, nv, idb in ((4, 1, 20_000, 3_000_000), (16, 1, 5_000, 4_000_000), (32, 1, 5_000, 5_000_000), (256, 1, 5_000, 2_000_000), (64, 4, 20_000, 1_000_000)):
        r = b._scene_record(ctx, n_chars, n_inst, nv, idb)
        r["scene"] = [n_chars, n_inst, nv]
        print(json.dumps({k: v for k, v in r.items() if k not in ("workload", "gpu_side_note", "skin_roofline", "parity")}), flush=True)
