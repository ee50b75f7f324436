#!/usr/bin/env python3
"""Experiment: lbs_skin_crowd on C3 -- launch period and the kernel's own duration (lbs.timing), exact and fused."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth
inst, verts, bones = 1000, int(os.environ.get("VERTS", "10000")), 64
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); ctx.set_option(k, int(v))
seed = synth.SEED_BASE + 3
mesh = synth.make_mesh(verts, bones, seed)
pal = synth.make_palette(bones, seed, n_instances=inst)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
nv = verts * inst
d_pal = ctx.to_device(pal)
outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
def launch(): ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, outs[0].ptr, outs[1].ptr, outs[2].ptr)
for exact in (1, 0):
    ctx.set_option("lbs.exact", exact)
    for _ in range(10): launch()
    per, ker = [], []
    for r in range(3):
        ctx.sync(); ctx.timer_begin()
        for _ in range(120): launch()
        per.append(ctx.timer_end() / 120 * 1e3)
        ctx.set_option("lbs.timing", 1); ctx.kernel_time()
        for _ in range(120): launch()
        us, n = ctx.kernel_time(); ctx.set_option("lbs.timing", 0)
        ker.append(us / n)
    print(json.dumps({"lib": os.path.basename(os.environ.get("FYX_LIB_PATH", "product")), "exact": exact, "period_us": round(float(np.median(per)), 2), "kernel_us": round(float(np.median(ker)), 2)}), flush=True)
