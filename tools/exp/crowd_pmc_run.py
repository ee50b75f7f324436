#!/usr/bin/env python3
"""A short serialized run of the C3 crowd skinning launch for rocprofv3 PMC passes (OPTS: library options, REPS: launches)."""
import os, sys
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
for _ in range(int(os.environ.get("REPS", "10"))):
    ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, outs[0].ptr, outs[1].ptr, outs[2].ptr)
    ctx.sync()
ctx.close()
