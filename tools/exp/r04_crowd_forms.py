#!/usr/bin/env python3
"""Round 4 experiment (time-boxed): two untried forms of the EXACT crowd kernel on C3, against the product in
the same process, interleaved (A B C ... A B C ..., so that drift hits all forms alike):
  form 1  two vertices per thread (1024-vertex tile), influences one by one (100 VGPRs)
  form 2  1024-thread workgroup, the product's arithmetic
  form 3  two vertices per thread, all twelve palette rows live (170 VGPRs: one workgroup per CU)
  form 4  1024-thread workgroup, influences one by one
Each with the launcher's run length and with runs of 8 instances.  kernel_us = the dispatch's own duration (lbs.timing).
Every form's output is compared with the product's, bit for bit (positions, normals, tangents of all 10 M vertices)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth
inst, verts, bones = 1000, 10_000, 64
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
seed = synth.SEED_BASE + 3
mesh = synth.make_mesh(verts, bones, seed)
pal = synth.make_palette(bones, seed, n_instances=inst)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
nv = verts * inst
d_pal = ctx.to_device(pal)
outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))


def launch():
    ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, outs[0].ptr, outs[1].ptr, outs[2].ptr)


def snapshot():
    ctx.sync()
    return [o.download(np.uint32, n) for o, n in zip(outs, (nv * 3, nv * 3, nv * 4))]


configs = [(0, 0), (0, 8), (1, 0), (1, 8), (1, 4), (2, 0), (2, 8), (3, 0), (3, 8), (4, 0), (4, 8)]
ref = None
same = {}
for form, ipb in configs:
    ctx.set_option("lbs.crowd_form", form)
    ctx.set_option("lbs.crowd_ipb", ipb)
    launch()
    got = snapshot()
    if ref is None:
        ref = got
    same[(form, ipb)] = bool(all(np.array_equal(a, b) for a, b in zip(ref, got)))
    del got
times = {c: [] for c in configs}
for rnd in range(4):
    for form, ipb in configs:
        ctx.set_option("lbs.crowd_form", form)
        ctx.set_option("lbs.crowd_ipb", ipb)
        for _ in range(8):
            launch()
        ctx.set_option("lbs.timing", 1)
        ctx.kernel_time()
        for _ in range(60):
            launch()
        us, n = ctx.kernel_time()
        ctx.set_option("lbs.timing", 0)
        times[(form, ipb)].append(us / n)
for c in configs:
    print(json.dumps({"form": c[0], "ipb": c[1] or "auto", "kernel_us_rounds": [round(t, 2) for t in times[c]], "kernel_us_median": round(float(np.median(times[c])), 2),
                      "frac_of_8TBps": round(404.696e6 / (float(np.median(times[c])) * 1e-6) / 8e12, 4), "bit_identical_to_product": same[c]}), flush=True)
ctx.close()
