#!/usr/bin/env python3
"""A short, serialized run for rocprofv3 PMC passes: N launches of the calibration stream
(known bytes: 60 MB read + 40 MB written) followed by N launches of the skinning kernel on the
C4 workload, rotating buffer sets.  Usage (separate passes, as MI355X_MICROARCH.md prescribes):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -- python tools/pmc_probe.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -- python tools/pmc_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fyrox_amd
from fyrox_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
SETS = 8
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
NV, NB, UNITS = 1_000_000, 256, 1_250_000
mesh = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
pal = synth.make_palette(NB, synth.SEED_BASE + 4)
d_pal = ctx.to_device(pal)
outs, srcs, dsts = [], [], []
for s in range(SETS):
    ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))
    srcs.append(ctx.to_device(np.full(UNITS * 12, np.float32(s + 1))))
    dsts.append(ctx.malloc(UNITS * 32))
for i in range(N):
    ctx.calib_stream_copy(srcs[i % SETS].ptr, dsts[i % SETS].ptr, UNITS)
ctx.sync()
for i in range(N):
    s = i % SETS
    ctx.lbs_skin_device(s, d_pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
ctx.sync()
# extended launches: 4 blend shapes -> SoA, vertex buffer in -> vertex buffer out (AnimatedVertex, 68 B), and the crowd kernel
L = synth.ANIMATED_VERTEX
storage, plane, w = synth.make_blend_shapes(NV, 4, synth.SEED_BASE + 4)
d_w = ctx.to_device(w)
aos = mesh.to_animated_vertex_aos()
vbs = []
for s in range(SETS):
    ctx.mesh_set_blend_shapes(s, storage, 4, plane)
    ctx.mesh_upload(100 + s, aos, NV, L["stride"], off_pos=L["off_pos"], off_normal=L["off_normal"], off_tangent=L["off_tangent"],
                    off_weights=L["off_weights"], off_indices=L["off_indices"])
    vbs.append(ctx.malloc(NV * L["stride"]))
for i in range(N):
    s = i % SETS
    ctx.lbs_skin_ex(s, d_pal.ptr, NB, 1, d_blend_shape_weights=d_w.ptr, n_blend_shapes=4, d_out_pos=outs[s][0].ptr,
                    d_out_normal=outs[s][1].ptr, d_out_tangent=outs[s][2].ptr)
ctx.sync()
for i in range(N):
    s = i % SETS
    ctx.lbs_skin_ex(100 + s, d_pal.ptr, NB, 1, d_out_vertices=vbs[s].ptr, out_stride=0)
ctx.sync()
# crowd: 100 instances x 10 k vertices / 64 bones (10 MB of palettes + mesh read, 40 MB written per launch)
cm = synth.make_mesh(10_000, 64, synth.SEED_BASE + 3)
cp = ctx.to_device(synth.make_palette(64, synth.SEED_BASE + 3, n_instances=100))
ctx.mesh_upload_soa(300, cm.pos, cm.weights, cm.indices, cm.normal, cm.tangent)
for i in range(N):
    s = i % SETS
    ctx.lbs_skin_device(300, cp.ptr, 64, 100, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
ctx.sync()
# batched launch: 64 meshes x 20 k vertices / 64 bones, one instance each, ONE lbs_skin_batch launch (128 MB algorithmic)
jobs = []
for k in range(64):
    bm = synth.make_mesh(20_000, 64, synth.SEED_BASE + 400 + k)
    ctx.mesh_upload_soa(400 + k, bm.pos, bm.weights, bm.indices, bm.normal, bm.tangent)
    bp = ctx.to_device(synth.make_palette(64, synth.SEED_BASE + 400 + k))
    bo = (ctx.malloc(20_000 * 12 + 64), ctx.malloc(20_000 * 12 + 64), ctx.malloc(20_000 * 16 + 64))
    jobs.append((400 + k, bp.ptr, 64, 1, bo[0].ptr, bo[1].ptr, bo[2].ptr))
    vbs.append(bp); vbs.extend(bo)     # keep alive
for i in range(N):
    ctx.lbs_skin_batch(jobs)
ctx.sync()
print("done")
