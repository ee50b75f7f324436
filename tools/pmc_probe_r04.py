#!/usr/bin/env python3
"""A short, serialized run for rocprofv3 PMC passes (round 4): N launches each, in this order, of
  0 the calibration stream copy (known bytes: 60 MB read + 40 MB written)
  1 C4 skinning, spatially coherent bone indices      2 C4 skinning, FULLY RANDOM bone indices (SURVEY 8(d) worst case)
  3 C3 crowd skinning (1000 x 10 k / 64), coherent    4 C3 crowd skinning, random indices
so that a per-dispatch counter file can be cut into five groups by dispatch order (the kernels of 1 / 2 and 3 / 4 have the same names).
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -- python tools/pmc_probe_r04.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fyrox_amd
from fyrox_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SETS = 6
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
NV, NB, UNITS = 1_000_000, 256, 1_250_000
pal = synth.make_palette(NB, synth.SEED_BASE + 4)
d_pal = ctx.to_device(pal)
meshes = {True: synth.make_mesh(NV, NB, synth.SEED_BASE + 4, coherent=True), False: synth.make_mesh(NV, NB, synth.SEED_BASE + 4, coherent=False)}
outs, srcs, dsts = [], [], []
for s in range(SETS):
    for k, coh in enumerate((True, False)):
        m = meshes[coh]
        ctx.mesh_upload_soa(10 * k + s, m.pos, m.weights, m.indices, m.normal, m.tangent)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))
    srcs.append(ctx.to_device(np.full(UNITS * 12, np.float32(s + 1))))
    dsts.append(ctx.malloc(UNITS * 32))
for i in range(N):
    ctx.calib_stream_copy(srcs[i % SETS].ptr, dsts[i % SETS].ptr, UNITS)
ctx.sync()
for k in range(2):
    for i in range(N):
        s = i % SETS
        ctx.lbs_skin_device(10 * k + s, d_pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
    ctx.sync()
inst = 1000
cp = ctx.to_device(synth.make_palette(64, synth.SEED_BASE + 3, n_instances=inst))
nv = 10_000 * inst
co = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
for k, coh in enumerate((True, False)):
    cm = synth.make_mesh(10_000, 64, synth.SEED_BASE + 3, coherent=coh)
    ctx.mesh_upload_soa(300 + k, cm.pos, cm.weights, cm.indices, cm.normal, cm.tangent)
    for i in range(N):
        ctx.lbs_skin_device(300 + k, cp.ptr, 64, inst, co[0].ptr, co[1].ptr, co[2].ptr)
    ctx.sync()
ctx.close()
