#!/usr/bin/env python3
"""Round 5: does evicting address translations between launches turn a FAST fused crowd launch into a slow one?  The fused crowd launch
(lbs.exact = 0, 1000 x 10 k / 64; per-dispatch events) alone, and with a launch between every two that walks over X MB of other memory
(a 1 M-vertex skinning launch rotating over k other sets = 100 k MB)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth

with fyrox_amd.Context(0) as ctx:
    ctx.set_option("lbs.streams", 1)
    ni, nvc, nbc = 1000, 10_000, 64
    meshc = synth.make_mesh(nvc, nbc, synth.SEED_BASE + 3)
    pals = ctx.to_device(np.concatenate([synth.make_palette(nbc, synth.SEED_BASE + 3 + (i % 7)) for i in range(ni)]))
    ctx.mesh_upload_soa(20, meshc.pos, meshc.weights, meshc.indices, meshc.normal, meshc.tangent)
    co = (ctx.malloc(ni * nvc * 12 + 64), ctx.malloc(ni * nvc * 12 + 64), ctx.malloc(ni * nvc * 16 + 64))
    nv, nb, K = 1_000_000, 256, 12
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 4)
    pal = ctx.to_device(synth.make_palette(nb, synth.SEED_BASE + 4))
    outs = []
    for m in range(K):
        ctx.mesh_upload_soa(100 + m, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        outs.append((ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)))
    res = {}
    for exact in (0, 1):
        ctx.set_option("lbs.exact", exact)
        for k in (0, 1, 4, 8, 12, 0):
            st = {"i": 0}

            def crowd():
                ctx.lbs_skin_device(20, pals.ptr, nbc, ni, co[0].ptr, co[1].ptr, co[2].ptr)

            def other():
                j = st["i"] % k
                st["i"] += 1
                o = outs[j]
                ctx.lbs_skin_device(100 + j, pal.ptr, nb, 1, o[0].ptr, o[1].ptr, o[2].ptr)
            for _ in range(10):
                crowd()
                if k:
                    other()
            ctx.sync()
            # time only the crowd launches: timing on for them, off for the others
            tot, cnt = 0.0, 0
            for _ in range(60):
                ctx.set_option("lbs.timing", 1)
                crowd()
                ctx.set_option("lbs.timing", 0)
                if k:
                    other()
            ctx.set_option("lbs.timing", 1)
            us, n = ctx.kernel_time()
            ctx.set_option("lbs.timing", 0)
            res.setdefault("exact" if exact else "fused", {}).setdefault(str(k), []).append(round(us / max(n, 1), 1))
    ctx.set_option("lbs.exact", 1)
    print(json.dumps({"crowd_kernel_us_with_a_launch_over_k_other_100MB_sets_between": res}))
