#!/usr/bin/env python3
"""Experiment: the static kernel lbs_skin with MANY short-lived workgroups (blocks_per_cu up to 64: one unit per wave)
against the resident-grid forms, lone launches on C4."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fyrox_amd
from fyrox_amd import synth
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
NV, NB, SETS = 1_000_000, 256, 8
mesh = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
pal = ctx.to_device(synth.make_palette(NB, synth.SEED_BASE + 4))
outs = []
for s in range(SETS):
    ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))
def launch(i):
    s = i % SETS
    ctx.lbs_skin_device(s, pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
def run(n):
    ctx.sync(); ctx.timer_begin()
    for i in range(n): launch(i)
    return ctx.timer_end() * 1e3 / n
cfgs = [("dyn b256", {"lbs.dyn": 1})]
for blk in (256, 512):
    for bpc in (4, 8, 16, 32, 61):
        for pf in (1, 0):
            cfgs.append((f"static b{blk} bpc{bpc} prefetch{pf}", {"lbs.dyn": 0, "lbs.block": blk, "lbs.blocks_per_cu": bpc, "lbs.prefetch": pf}))
res = {}
for rnd in range(3):
    for name, o in cfgs:
        for k, v in o.items(): ctx.set_option(k, v)
        run(20)
        res.setdefault(name, []).append(run(200))
for name, _ in cfgs:
    print(json.dumps({"cfg": name, "lone_us": round(float(np.median(res[name])), 2)}), flush=True)
