#!/usr/bin/env python3
"""Experiment: per-wave timeline of lbs_skin_dyn WITH the hardware position of every wave (XCC, SE, CU), dumped raw.
    python tools/exp/dyn_where.py 1024 0 gpurun_out/where_1024.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fyrox_amd
from fyrox_amd import synth
blk, kn, out = int(sys.argv[1]), int(sys.argv[2], 0), sys.argv[3]
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
for kv in sys.argv[4:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
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
ctx.set_option("lbs.dyn", 1); ctx.set_option("lbs.dyn_block", blk); ctx.set_option("lbs.dyn_knobs", kn)
for i in range(40): launch(i)
ctx.sync()
ctx.set_option("lbs.probe", 1)
all_ = []
for i in range(16):
    launch(i); launch(i + 1); ctx.sync()
    buf = np.zeros((4096, 4), np.uint64)
    ctx._check(ctx._l.fyx_debug_read_probe(ctx._h, buf.ctypes.data, 4096))
    all_.append(buf.copy())
np.savez_compressed(out, probe=np.stack(all_))
