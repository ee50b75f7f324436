#!/usr/bin/env python3
"""bench.py's vertex-buffer record alone (lbs_skin_aos: vertex buffer in -> vertex buffer out, plain and with 4 blend shapes)."""
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
    for _ in range(2):
        r = b._vertex_buffer_record(ctx)
        print(json.dumps({k: {"period_us": r[k]["launch_period_us_one_stream"], "frac": r[k]["roofline"]["frac"], "frac_two_streams": r[k]["roofline"]["frac_two_streams"],
                              "bit_exact": r[k]["parity"]["bit_exact"]} for k in ("plain", "with_4_blend_shapes")}), flush=True)
