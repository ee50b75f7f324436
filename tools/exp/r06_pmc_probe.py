#!/usr/bin/env python3
"""Round 6: launches of the lone C4 kernel rotating over K buffer sets, for a rocprofv3 --pmc pass (the footprint knee: VERDICT r5 item 2a).
    python tools/exp/r06_pmc_probe.py K [launches] [option=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth

K = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 48
with fyrox_amd.Context(0) as ctx:
    ctx.set_option("lbs.streams", 1)
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    nv, nb = 1_000_000, 256
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 4)
    pal = ctx.to_device(synth.make_palette(nb, synth.SEED_BASE + 4))
    outs = []
    for m in range(K):
        ctx.mesh_upload_soa(1 + m, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        outs.append((ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)))
    for i in range(N):
        o = outs[i % K]
        ctx.lbs_skin_device(1 + i % K, pal.ptr, nb, 1, o[0].ptr, o[1].ptr, o[2].ptr)
    ctx.sync()
