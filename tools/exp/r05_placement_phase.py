#!/usr/bin/env python3
"""Round 5: the fused crowd launch (lbs.exact = 0, 1000 x 10 k / 64) with its three output streams carved from ONE block at varied
gaps, and the same layout in a second block: is fast / slow a matter of the streams' relative phase, or of the block?"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth


def kernel_us(ctx, launch, n=60, warm=10):
    for _ in range(warm):
        launch()
    ctx.sync()
    ctx.set_option("lbs.timing", 1)
    ctx.kernel_time()
    for _ in range(n):
        launch()
    us, cnt = ctx.kernel_time()
    ctx.set_option("lbs.timing", 0)
    return us / max(cnt, 1)


with fyrox_amd.Context(0) as ctx:
    ctx.set_option("lbs.streams", 1)
    ni, nv, nb = 1000, 10_000, 64
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 3)
    pals = ctx.to_device(np.concatenate([synth.make_palette(nb, synth.SEED_BASE + 3 + (i % 7)) for i in range(ni)]))
    ctx.mesh_upload_soa(20, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    ctx.set_option("lbs.exact", 0)
    b12, b16 = ni * nv * 12, ni * nv * 16
    gaps = [0, 4096, 65536, 1 << 20, 3 << 20, 16 << 20]
    blocks = [ctx.malloc(2 * b12 + b16 + 3 * (16 << 20) + (1 << 20)) for _ in range(3)]
    for bi, blk in enumerate(blocks):
        for shift in (0, 1 << 20):
            row = []
            for g1 in gaps:
                for g2 in gaps:
                    p = blk.ptr + shift
                    n = p + b12 + g1
                    t = n + b12 + g2
                    us = kernel_us(ctx, lambda: ctx.lbs_skin_device(20, pals.ptr, nb, ni, p, n, t))
                    row.append(round(us, 1))
            print(json.dumps({"block": bi, "base": hex(blk.ptr), "shift": shift, "gaps": gaps, "kernel_us_by_gap1_gap2": row}), flush=True)
    # three separate allocations, as callers usually make them
    for rep in range(4):
        o = [ctx.malloc(b12 + 64), ctx.malloc(b12 + 64), ctx.malloc(b16 + 64)]
        us = kernel_us(ctx, lambda: ctx.lbs_skin_device(20, pals.ptr, nb, ni, o[0].ptr, o[1].ptr, o[2].ptr))
        print(json.dumps({"separate": [hex(b.ptr) for b in o], "kernel_us": round(us, 1)}), flush=True)
        if rep % 2 == 0:
            for b in o:
                b.free()
