#!/usr/bin/env python3
"""Experiment: does the crowd kernel's (fused, store-bound) duration depend on WHERE its three output streams were allocated?
Several allocations of the same 120 / 120 / 160 MB in one process, each timed with back-to-back launches; the buffers' device
addresses are printed modulo 4 KB ... 1 GB."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth
inst, verts, bones = 1000, 10000, 64
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
seed = synth.SEED_BASE + 3
mesh = synth.make_mesh(verts, bones, seed)
pal = synth.make_palette(bones, seed, n_instances=inst)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
nv = verts * inst
d_pal = ctx.to_device(pal)
keep = []
for trial in range(8):
    pad = [ctx.malloc(int(x)) for x in ([], [1 << 20], [3 << 20, 5 << 20], [64 << 20], [], [7 << 20], [1 << 30], [])[trial]]
    outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
    keep += pad
    def launch(): ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, outs[0].ptr, outs[1].ptr, outs[2].ptr)
    row = {"trial": trial, "addr_mod_2MB": [o.ptr % (2 << 20) for o in outs], "addr_GB": [round(o.ptr / 2**30, 3) for o in outs]}
    for exact in (0, 1):
        ctx.set_option("lbs.exact", exact)
        for _ in range(10): launch()
        ctx.set_option("lbs.timing", 1); ctx.kernel_time()
        for _ in range(80): launch()
        us, n = ctx.kernel_time(); ctx.set_option("lbs.timing", 0)
        row["fused_us" if exact == 0 else "exact_us"] = round(us / n, 2)
    print(json.dumps(row), flush=True)
    if trial % 2 == 0:
        for o in outs: o.free()
    else:
        keep += list(outs)
ctx.close()
