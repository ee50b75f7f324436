#!/usr/bin/env python3
"""Sweep the instanced-skinning options on the C3 crowd (skinning only, palettes resident):
streaming kernel (lbs.crowd=0) vs crowd kernel (lbs.crowd=1) x tile size x instances per run.
One JSON line per configuration."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--instances", type=int, default=1000)
ap.add_argument("--verts", type=int, default=10_000)
ap.add_argument("--bones", type=int, default=64)
ap.add_argument("--reps", type=int, default=100)
args = ap.parse_args()

ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
seed = synth.SEED_BASE + 3
mesh = synth.make_mesh(args.verts, args.bones, seed)
pal = synth.make_palette(args.bones, seed, n_instances=args.instances)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
nv = args.verts * args.instances
d_pal = ctx.to_device(pal)
d_pos, d_nrm, d_tan = ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)
unique = args.verts * 60 + args.instances * args.bones * 64 + nv * 40


def run(label, **opts):
    for k, v in opts.items():
        ctx.set_option("lbs." + k, v)
    for _ in range(5):
        ctx.lbs_skin_device(3, d_pal.ptr, args.bones, args.instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(args.reps):
        ctx.lbs_skin_device(3, d_pal.ptr, args.bones, args.instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
    ms = ctx.timer_end() / args.reps
    print(json.dumps({"cfg": label, **opts, "ms": round(ms, 5), "verts_per_s": nv / (ms * 1e-3),
                      "unique_GBps": unique / (ms * 1e-3) / 1e9}), flush=True)


run("stream", crowd=0)
for exact in (1, 0):
    for ipb in (0, 2, 4, 8, 16, 32, 64):
        run("crowd", crowd=1, crowd_ipb=ipb, exact=exact)
ctx.close()
