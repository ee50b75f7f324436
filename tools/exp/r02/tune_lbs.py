#!/usr/bin/env python3
"""Sweep the skinning kernel's tuning knobs on the C4 workload (1 M verts / 256 bones) and print
the average launch time of each variant (HIP events on the launch stream, rotating buffer sets).
Run on the GPU box:  python tools/tune_lbs.py [--quick] > gpurun_out/tune.json"""
import argparse
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fyrox_amd
from fyrox_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--verts", type=int, default=1_000_000)
ap.add_argument("--bones", type=int, default=256)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--random-bones", action="store_true")
ap.add_argument("--blocks", default="256,512,1024")
ap.add_argument("--bpcu", default="1,2,3,4,5,6,8,12,16")
ap.add_argument("--prefetch", default="1,0")
ap.add_argument("--nt", default="1")
ap.add_argument("--exact", default="1,0")
ap.add_argument("--streams", default="2,1")
args = ap.parse_args()

ctx = fyrox_amd.Context(0)
mesh = synth.make_mesh(args.verts, args.bones, synth.SEED_BASE + 4, coherent=not args.random_bones)
pal = synth.make_palette(args.bones, synth.SEED_BASE + 4)
d_pal = ctx.to_device(pal)
nv = mesh.n_verts
outs = []
for s in range(args.sets):
    ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs.append((ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)))


def run(steps):
    ctx.timer_begin()
    for i in range(steps):
        s = i % args.sets
        ctx.lbs_skin_device(s, d_pal.ptr, args.bones, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
    return ctx.timer_end() * 1e3 / steps  # us per launch


ints = lambda s: [int(x) for x in s.split(",")]
variants = list(itertools.product(ints(args.exact), ints(args.prefetch), ints(args.nt), ints(args.blocks), ints(args.bpcu), ints(args.streams)))
results = {}
for r in range(args.rounds):            # interleaved rounds: variant order repeated, min/median reported
    for v in variants:
        exact, prefetch, nt, block, bpcu, streams = v
        if block * bpcu > 2048 * 2:      # more than 64 waves/CU requested: pointless
            continue
        for k, val in (("lbs.exact", exact), ("lbs.prefetch", prefetch), ("lbs.nt", nt), ("lbs.block", block), ("lbs.blocks_per_cu", bpcu), ("lbs.streams", streams)):
            ctx.set_option(k, val)
        run(20)
        results.setdefault(v, []).append(run(args.steps))

rows = []
for v, ts in results.items():
    us = float(np.median(ts))
    rows.append({"exact": v[0], "prefetch": v[1], "nt": v[2], "block": v[3], "blocks_per_cu": v[4], "streams": v[5],
                 "us_median": us, "us_min": float(min(ts)), "GBps": 100.0 * nv / us / 1e3,
                 "frac_of_8TBps": 100.0 * nv / us / 1e3 / 8000.0})
rows.sort(key=lambda r: r["us_median"])
print(json.dumps({"verts": nv, "bones": args.bones, "random_bones": args.random_bones, "rows": rows}, indent=1))
for r in rows[:12]:
    print("#", r, file=sys.stderr)
