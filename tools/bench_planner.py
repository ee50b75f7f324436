#!/usr/bin/env python3
"""Host control plane alone (no GPU: a control-only context): microseconds per frame to plan N instances of the C5
blend-tree machine (4 clips) -- from scratch (the memo defeated by a no-op setter before every frame) and with the
fold-program memo.  One JSON line.

    python tools/bench_planner.py [--instances 1000]
"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fyrox_amd
from fyrox_amd import anim as A, synth

ap = argparse.ArgumentParser()
ap.add_argument("--instances", type=int, default=1000)
ap.add_argument("--frames", type=int, default=300)
args = ap.parse_args()
ctx = fyrox_amd.Context(control_only=True)
N, seed = args.instances, synth.SEED_BASE + 3
rig = synth.make_rig(64, seed)
A.create_rig(ctx, 1, rig)
an = A.Animator(ctx, 1, 1, rig, N)
for c in range(4):
    td, tgt = synth.make_clip(64, seed, clip=c)
    A.upload_tracks_data(ctx, 10 + c, td)
    an.add_animation(10 + c, tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
an.set_machine(synth.make_c5_machine())
for i in range(N):
    for c in range(4):
        an.set_time_position(c, (i * 0.37 + c * 0.11) % 1.0, instance=i)
plan, setloop = ctx._l.fyx_animator_plan, ctx._l.fyx_animation_set_loop
n, dt = ctypes.c_uint32(), ctypes.c_float(1 / 60)


def run(defeat: bool) -> float:
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(args.frames):
            if defeat:
                setloop(ctx._h, an.id, 0, 0xFFFFFFFF, 1)      # same value: only invalidates the memo
            plan(ctx._h, an.id, 1, dt, None, None, None, None, 0, ctypes.byref(n))
        best = min(best, (time.perf_counter() - t0) / args.frames)
    return best * 1e6


for _ in range(50):
    plan(ctx._h, an.id, 1, dt, None, None, None, None, 0, ctypes.byref(n))
scratch, memo = run(True), run(False)
print(json.dumps({"instances": N, "ops_per_frame": n.value, "plan_us_from_scratch": round(scratch, 1), "plan_us_with_memo": round(memo, 1),
                  "ns_per_instance_from_scratch": round(scratch * 1e3 / N, 1), "ns_per_instance_with_memo": round(memo * 1e3 / N, 1)}))
ctx.close()
