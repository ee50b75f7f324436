#!/usr/bin/env python3
"""Single characters (BASELINE configs 2 and 5) alone: bench.py's own chain record for C2 and C5 in a fresh process, so that a
kernel trace of this command (tools/profile.sh) shows what one character's frame consists of.  One JSON line.  GPU only."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("fyx_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
argv, sys.argv = sys.argv, ["bench.py"]
spec.loader.exec_module(bench)
sys.argv = argv
import anim_cases as cases
import fyrox_amd
from fyrox_amd import synth

ctx = fyrox_amd.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
ctx.set_option("lbs.streams", 1)
rig2 = synth.make_rig(64, synth.SEED_BASE + 2)
td, tgt = synth.make_clip(64, synth.SEED_BASE + 2, 0)
c2 = cases.Scenario("c2", rig2, [td], [cases.AnimSpec(0, tgt)], None, n_frames=20)
out = {"c2": bench._chain_record(ctx, "C2: one character, 50k verts / 64 bones / 1 clip", c2, synth.make_mesh(50_000, 64, synth.SEED_BASE + 2), 1, 400, False, [0]),
       "c5": bench._chain_record(ctx, "C5: Machine 4-clip blend tree -> palette -> 100k-vert LBS", cases.c5_blend_tree(n_bones=64),
                                 synth.make_mesh(100_000, 64, synth.SEED_BASE + 5), 1, 400, False, [0])}
print(json.dumps({k: {kk: v[kk] for kk in ("frame_ms", "frame_ms_one_stream", "pose_ms", "skin_ms")} for k, v in out.items()}))
ctx.close()
