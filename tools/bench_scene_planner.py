#!/usr/bin/env python3
"""Host control plane of a SCENE alone (no GPU: a control-only context): microseconds per frame for fyx_scene_plan over
`--characters` animators of one instance each (every one its own rig-sized C5 blend-tree machine), with the planner
threads (`--threads`, option anim.threads) and on the calling thread only.  One JSON line.

    python tools/bench_scene_planner.py [--characters 256]
"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth

ap = argparse.ArgumentParser()
ap.add_argument("--characters", type=int, default=256)
ap.add_argument("--instances", type=int, default=1)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--threads", type=int, default=8)
args = ap.parse_args()
ctx = fyrox_amd.Context(control_only=True)
seed = synth.SEED_BASE + 3
rig = synth.make_rig(64, seed)
A.create_rig(ctx, 1, rig)
tgts = []
for c in range(4):
    td, tgt = synth.make_clip(64, seed, clip=c)
    A.upload_tracks_data(ctx, 10 + c, td)
    tgts.append(tgt)
ids = []
for k in range(args.characters):
    an = A.Animator(ctx, 100 + k, 1, rig, args.instances)
    for c in range(4):
        an.add_animation(10 + c, tgts[c], time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
    an.set_machine(synth.make_c5_machine())
    for c in range(4):
        an.set_time_position(c, (k * 0.37 + c * 0.11) % 1.0)
    ids.append(100 + k)
arr = np.asarray(ids, np.uint64)
plan = ctx._l.fyx_scene_plan
p = arr.ctypes.data_as(ctypes.c_void_p)
dt = ctypes.c_float(1 / 60)


def run() -> float:
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(args.frames):
            rc = plan(ctx._h, p, len(ids), dt)
        assert rc == 0
        best = min(best, (time.perf_counter() - t0) / args.frames)
    return best * 1e6


out = {"characters": args.characters, "instances_each": args.instances}
for th in (1, args.threads):
    ctx.set_option("anim.threads", th)
    for _ in range(50):
        plan(ctx._h, p, len(ids), dt)
    us = run()
    out[f"plan_us_threads_{th}"] = round(us, 1)
    out[f"ns_per_character_threads_{th}"] = round(us * 1e3 / args.characters, 1)
print(json.dumps(out))
ctx.close()
