#!/usr/bin/env python3
"""Calibration: the pure-stream ceiling (fyx_calib_stream_copy, 60 MB read + 40 MB written per
launch, same rotating-buffer protocol) next to the skinning kernel, in one process.
  python tools/calib.py [--steps 400]
Under rocprofv3 --kernel-trace --stats this also gives pure kernel durations (no launch gaps)."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fyrox_amd
from fyrox_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--variants", default="512x2,512x4,512x8")
args = ap.parse_args()

ctx = fyrox_amd.Context(0)
NV, NB = 1_000_000, 256
UNITS = 1_250_000
mesh = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
pal = synth.make_palette(NB, synth.SEED_BASE + 4)
d_pal = ctx.to_device(pal)
outs, srcs, dsts = [], [], []
for s in range(args.sets):
    ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))
    srcs.append(ctx.to_device(np.zeros(UNITS * 12, np.float32) + np.float32(s)))
    dsts.append(ctx.malloc(UNITS * 32))


def time_it(fn, steps):
    for i in range(20):
        fn(i)
    ctx.timer_begin()
    for i in range(steps):
        fn(i)
    return ctx.timer_end() * 1e3 / steps


def lbs(i):
    s = i % args.sets
    ctx.lbs_skin_device(s, d_pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)


def copy(i):
    s = i % args.sets
    ctx.calib_stream_copy(srcs[s].ptr, dsts[s].ptr, UNITS)


res = {"copy": {}, "lbs": {}}
for r in range(args.rounds):
    for bpcu in (2, 4, 8, 16, 32):
        ctx.set_option("lbs.blocks_per_cu", bpcu)
        res["copy"].setdefault(f"256x{bpcu}", []).append(time_it(copy, args.steps))
    for v in args.variants.split(","):
        b, g = v.split("x")       # (the workgroup size is fixed at 512 since round 3; only the grid multiple varies)
        ctx.set_option("lbs.blocks_per_cu", int(g))
        for exact in (1, 0):
            ctx.set_option("lbs.exact", exact)
            res["lbs"].setdefault(f"{v} exact={exact}", []).append(time_it(lbs, args.steps))
out = {k: {n: {"us": float(np.median(t)), "GBps": 100e6 / np.median(t) / 1e3} for n, t in d.items()} for k, d in res.items()}
print(json.dumps(out, indent=1))
