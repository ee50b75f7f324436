#!/usr/bin/env python3
"""Mid-size single meshes: fyx_lbs_skin_device (lbs_skin below 512 K vertices, lbs_skin_dyn from there) against a one-job fyx_lbs_skin_batch
(lbs_skin_batch_dyn), back-to-back launches on one stream, GPU timer.  One JSON line per size."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import fyrox_amd
from fyrox_amd import synth
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
for nv, nb, ni in ((20_000, 64, 1), (50_000, 64, 1), (100_000, 64, 1), (200_000, 128, 1), (400_000, 256, 1), (1_000_000, 256, 1), (50_000, 64, 2), (50_000, 64, 3), (200_000, 64, 3)):
    m = synth.make_mesh(nv, nb, 7)
    pal = ctx.to_device(synth.make_palette(nb, 7, n_instances=ni))
    ctx.mesh_upload_soa(5, m.pos, m.weights, m.indices, m.normal, m.tangent)
    outs = [ctx.malloc(nv * ni * w * 4 + 64) for w in (3, 3, 4)]
    job = [(5, pal.ptr, nb, ni, outs[0].ptr, outs[1].ptr, outs[2].ptr)]
    res = {}
    for name, fn in (("device", lambda: ctx.lbs_skin_device(5, pal.ptr, nb, ni, outs[0].ptr, outs[1].ptr, outs[2].ptr)), ("batch", lambda: ctx.lbs_skin_batch(job))):
        for _ in range(20):
            fn()
        best = 1e9
        for _ in range(3):
            ctx.sync(); ctx.timer_begin()
            for _ in range(200):
                fn()
            best = min(best, ctx.timer_end() / 200)
        res[name + "_us"] = round(best * 1e3, 3)
    res.update(verts=nv, bones=nb, instances=ni, GBps_device=round(nv * ni * 100 / res["device_us"] / 1e3, 1), GBps_batch=round(nv * ni * 100 / res["batch_us"] / 1e3, 1))
    print(json.dumps(res), flush=True)
    for o in outs: o.free()
    pal.free(); ctx.mesh_free(5)
