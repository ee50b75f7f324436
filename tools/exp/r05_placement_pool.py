#!/usr/bin/env python3
"""Round 5, VERDICT item 4c: does choosing WHERE a large output lies by measurement (fyx_malloc_probed: the fastest-filling of K candidate
blocks) move the store-bound kernels?  C4's lone launch (lbs_skin_dyn, 1 M vertices / 256 bones) over 6 output sets, and the fused crowd
launch (lbs.exact = 0, 1000 x 10 k), outputs from fyx_malloc against fyx_malloc_probed.  One JSON line per leg.

    python tools/exp/r05_placement_pool.py [candidates]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd                      # noqa: E402
from fyrox_amd import synth           # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def kernel_us(ctx, launch, n=200, warm=30):
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


def main():
    with fyrox_amd.Context(0) as ctx:
        ctx.set_option("lbs.streams", 1)
        # ---- C4: as bench.py runs it -- six rotating sets of inputs AND outputs (600 MB >> the 256 MiB Infinity Cache) ----
        nv, nb = 1_000_000, 256
        mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 4)
        pal = ctx.to_device(synth.make_palette(nb, synth.SEED_BASE + 4))
        for m in range(6):
            ctx.mesh_upload_soa(1 + m, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        for probe in (0, K, 0, K):
            sets, fills = [], []
            for _ in range(6):
                o = (ctx.malloc(nv * 12 + 64, probe), ctx.malloc(nv * 12 + 64, probe), ctx.malloc(nv * 16 + 64, probe))
                sets.append(o)
                fills.append([b.fill_us for b in o])
            st = {"k": 0}

            def rot():
                k = st["k"] % 6
                st["k"] += 1
                o = sets[k]
                ctx.lbs_skin_device(1 + k, pal.ptr, nb, 1, o[0].ptr, o[1].ptr, o[2].ptr)
            us = [kernel_us(ctx, rot, n=600) for _ in range(3)]
            print(json.dumps({"leg": "c4_lone_launch_rotating_6_sets", "candidates": probe, "kernel_us": [round(x, 2) for x in us],
                              "frac_of_8TBps": round(100e6 / (float(np.median(us)) * 1e-6) / 8e12, 3),
                              "fill_us_of_the_candidates": fills if probe else None}), flush=True)
            for o in sets:
                for b in o:
                    b.free()
        for m in range(6):
            ctx.mesh_free(1 + m)
        # ---- fused crowd ----
        ni, nvc, nbc = 1000, 10_000, 64
        meshc = synth.make_mesh(nvc, nbc, synth.SEED_BASE + 3)
        pals = ctx.to_device(np.concatenate([synth.make_palette(nbc, synth.SEED_BASE + 3 + (i % 7)) for i in range(ni)]))
        ctx.mesh_upload_soa(20, meshc.pos, meshc.weights, meshc.indices, meshc.normal, meshc.tangent)
        ctx.set_option("lbs.exact", 0)
        unique = nvc * 60 + ni * nbc * 64 + ni * nvc * 40
        for rep_ in range(3):      # the three streams in ONE slab, candidates probed with the skinning store pattern
            (p_, n_, t_), fills = ctx.malloc_skinned_probed(ni * nvc, K)
            us = kernel_us(ctx, lambda: ctx.lbs_skin_device(20, pals.ptr, nbc, ni, p_, n_, t_), n=100, warm=20)
            print(json.dumps({"leg": "c3_fused_slab_probed", "candidates": K, "kernel_us": round(us, 2), "frac_of_8TBps": round(unique / (us * 1e-6) / 8e12, 3),
                              "fill_us_of_the_candidates": fills}), flush=True)
            ctx.free_ptr(p_)
        for probe in (0, K, 0, K):
            o = (ctx.malloc(ni * nvc * 12 + 64, probe), ctx.malloc(ni * nvc * 12 + 64, probe), ctx.malloc(ni * nvc * 16 + 64, probe))
            us = kernel_us(ctx, lambda: ctx.lbs_skin_device(20, pals.ptr, nbc, ni, o[0].ptr, o[1].ptr, o[2].ptr), n=100, warm=20)
            print(json.dumps({"leg": "c3_fused", "candidates": probe, "kernel_us": round(us, 2), "frac_of_8TBps": round(unique / (us * 1e-6) / 8e12, 3),
                              "fill_us_of_the_candidates": [b.fill_us for b in o] if probe else None}), flush=True)
            for b in o:
                b.free()
        ctx.set_option("lbs.exact", 1)


if __name__ == "__main__":
    main()
