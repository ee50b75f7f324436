#!/usr/bin/env python3
"""Experiment: where a wave of lbs_skin_crowd spends its cycles on C3 (probe variant, lbs.crowd_stage=4): per wave the sums over its
instances of {waiting at the barrier, barrier -> results (LDS gathers + arithmetic), results -> loop end (stores, palette commit)}
and its lifetime, in shader clocks (s_memtime)."""
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
outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
def launch(): ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, outs[0].ptr, outs[1].ptr, outs[2].ptr)
ctx.set_option("lbs.probe", 1)
for exact in (1, 0):
    ctx.set_option("lbs.exact", exact)
    for ipb in (16, 8):
        ctx.set_option("lbs.crowd_ipb", ipb)
        ctx.set_option("lbs.crowd_stage", 0)
        for _ in range(5): launch()
        ctx.set_option("lbs.crowd_stage", 4)
        tiles, chunks = (verts + 511) // 512, (inst + ipb - 1) // ipb
        n_waves = tiles * chunks * 8
        for _ in range(3): launch()
        ctx.sync(); ctx.timer_begin(); launch(); ms = ctx.timer_end()
        buf = np.zeros((n_waves, 4), np.uint64)
        ctx._check(ctx._l.fyx_debug_read_probe(ctx._h, buf.ctypes.data, n_waves))
        b = buf.astype(np.float64)
        w = np.arange(n_waves) % 8
        full = b[:, 1] > 0
        row = {"exact": exact, "ipb": ipb, "probed_launch_us": round(ms * 1e3, 1), "waves": int(n_waves),
               "per_instance_cycles_all_waves": {k: round(float(b[full, i].sum() / (full.sum() * ipb)), 1) for i, k in enumerate(("barrier", "math", "tail", "palette_wait"))},
               
               "by_wave_slot_per_instance": {k: [round(float(b[full & (w == s), i].mean() / ipb), 1) for s in range(8)] for i, k in enumerate(("barrier", "math", "tail", "palette_wait"))}}
        print(json.dumps(row), flush=True)
ctx.close()
