#!/usr/bin/env python3
"""Experiment: lone-launch time of the default single-instance skinning kernel and of the no-math calibration copy at
several mesh sizes -- the fixed cost of a launch (ramp + tail) vs its streaming rate.  One JSON line per size."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fyrox_amd
from fyrox_amd import synth
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, int(v, 0))
NB = 256
pal = ctx.to_device(synth.make_palette(NB, synth.SEED_BASE + 4))
base = synth.make_mesh(1_000_000, NB, synth.SEED_BASE + 4)
for mult in (1, 2, 4, 8):
    nv = 1_000_000 * mult
    rep = lambda a: np.concatenate([a] * mult)
    sets = max(2, 8 // mult)
    outs = []
    for s in range(sets):
        ctx.mesh_upload_soa(s, rep(base.pos), rep(base.weights), rep(base.indices), rep(base.normal), rep(base.tangent))
        outs.append((ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)))
    def run(steps):
        ctx.timer_begin()
        for i in range(steps):
            s = i % sets
            ctx.lbs_skin_device(s, pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
        return ctx.timer_end() * 1e3 / steps
    run(20)
    ts = [run(max(40, 300 // mult)) for _ in range(3)]
    units = nv * 100 // 80
    srcs = [ctx.malloc(units * 48) for _ in range(sets)]
    dsts = [ctx.malloc(units * 32) for _ in range(sets)]
    def runc(steps):
        ctx.timer_begin()
        for i in range(steps):
            ctx.calib_stream_copy(srcs[i % sets].ptr, dsts[i % sets].ptr, units)
        return ctx.timer_end() * 1e3 / steps
    runc(10)
    cs = [runc(max(40, 300 // mult)) for _ in range(3)]
    print(json.dumps({"verts": nv, "lbs_us": float(np.median(ts)), "copy_us": float(np.median(cs)),
                      "lbs_TBps": nv * 100 / np.median(ts) / 1e6, "copy_TBps": nv * 100 / np.median(cs) / 1e6}), flush=True)
    for b in srcs + dsts:
        b.free()
    for o in outs:
        for b in o: b.free()
    for s in range(sets):
        ctx.mesh_free(s)
