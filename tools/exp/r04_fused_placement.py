#!/usr/bin/env python3
"""Round 4 experiment: the fused crowd launch (lbs.exact = 0; it IS its 400 MB of stores) runs at 60 us on some
allocations of the three output streams and 78 us on others.  Is there a predictor in the ADDRESSES?
 (a) eight separate allocations of the three streams (what callers do): addresses and kernel time;
 (b) ONE slab, the three streams at chosen relative offsets: pos at 0, normals and tangents behind it with paddings that shift
     their phase against pos (the same vertex is written to all three at the same moment: 12 v, 12 v, 16 v bytes into them).
kernel_us = the dispatch's own duration (lbs.timing), 60 launches after 10 warm-ups.  One JSON line per configuration."""
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
B = (nv * 12, nv * 12, nv * 16)


def timed(p, n, t, exact):
    ctx.set_option("lbs.exact", exact)
    for _ in range(10):
        ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, p, n, t)
    ctx.set_option("lbs.timing", 1)
    ctx.kernel_time()
    for _ in range(60):
        ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, p, n, t)
    us, k = ctx.kernel_time()
    ctx.set_option("lbs.timing", 0)
    return us / k


def describe(p, n, t):
    return {"pos": hex(p), "nrm": hex(n), "tan": hex(t), "nrm_minus_pos_mod_2M": (n - p) % (2 << 20), "tan_minus_pos_mod_2M": (t - p) % (2 << 20),
            "pos_mod_1G": p % (1 << 30), "nrm_minus_pos_mod_64K": (n - p) % 65536, "tan_minus_pos_mod_64K": (t - p) % 65536}


which = os.environ.get("WHICH", "slabs")
if which == "pmc":
    # eight separate allocations held at once, three fused launches each in allocation order (for a counter pass), then their times
    held = [[ctx.malloc(b + 64) for b in B] for _ in range(8)]
    ctx.set_option("lbs.exact", 0)
    for bufs in held:
        for _ in range(3):
            ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr)
        ctx.sync()
    for k, bufs in enumerate(held):
        print(json.dumps({"kind": "separate allocations", "k": k, "fused_us": round(timed(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, 0), 2), "pos": hex(bufs[0].ptr)}), flush=True)
    ctx.set_option("lbs.exact", 1)
    ctx.close()
    sys.exit(0)
if which == "slabs":
    # (c) is ONE allocation for the three streams reliably fast?  eight slabs held at once, then eight separate sets held at once,
    # then slabs again -- all in this process
    for rnd in range(2):
        slabs = [ctx.malloc(sum(B) + (4 << 20)) for _ in range(8)]
        for k, sl in enumerate(slabs):
            p = (sl.ptr + 255) // 256 * 256
            n = p + B[0]
            t = n + B[1]
            print(json.dumps({"kind": "one allocation for the three streams", "round": rnd, "k": k, "fused_us": round(timed(p, n, t, 0), 2), "exact_us": round(timed(p, n, t, 1), 2),
                              "base": hex(sl.ptr)}), flush=True)
        for sl in slabs:
            sl.free()
        held = []
        for k in range(8):
            bufs = [ctx.malloc(b + 64) for b in B]
            held.append(bufs)
            p, n, t = (b.ptr for b in bufs)
            print(json.dumps({"kind": "separate allocations", "round": rnd, "k": k, "fused_us": round(timed(p, n, t, 0), 2), "exact_us": round(timed(p, n, t, 1), 2), **describe(p, n, t)}), flush=True)
        for bufs in held:
            for b in bufs:
                b.free()
    ctx.set_option("lbs.exact", 1)
    ctx.close()
    sys.exit(0)
held = []
for k in range(8):
    bufs = [ctx.malloc(b + 64) for b in B]
    held.append(bufs)
    p, n, t = (b.ptr for b in bufs)
    print(json.dumps({"kind": "separate allocations", "k": k, "fused_us": round(timed(p, n, t, 0), 2), "exact_us": round(timed(p, n, t, 1), 2), **describe(p, n, t)}), flush=True)
for bufs in held:
    for b in bufs:
        b.free()
slab = ctx.malloc(sum(B) + (64 << 20))
base = (slab.ptr + (2 << 20) - 1) // (2 << 20) * (2 << 20)
pads = [0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 65536 + 4096, 3 * 4096 + 128]
for d1 in pads:
    for d2 in (0, 2 * d1 + 512, 8192 + 384):
        p = base
        n = base + B[0] + d1
        n = (n + 15) // 16 * 16
        t = n + B[1] + d2
        t = (t + 15) // 16 * 16
        print(json.dumps({"kind": "one slab", "pad_nrm": d1, "pad_tan": d2, "fused_us": round(timed(p, n, t, 0), 2), **describe(p, n, t)}), flush=True)
ctx.set_option("lbs.exact", 1)
ctx.close()
