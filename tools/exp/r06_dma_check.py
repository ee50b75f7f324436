#!/usr/bin/env python3
"""Round 6: lbs_skin_dma (lbs.dyn = 2) against the oracle, bit for bit: sizes with ragged ends, bone counts, every output set, projective
palette; then the same launches as lbs.dyn = 1 for the timing A/B (tools/exp/r06_sets_sweep.py lbs.dyn 1,2)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth
import oracle

bad = []
with fyrox_amd.Context(0) as ctx:
    ctx.set_option("lbs.streams", 1)
    for nv, nb in ((1_000_000, 256), (1_048_576, 200), (1_000_003, 5), (530_001, 64), (2_000_001, 256)):
        m = synth.make_mesh(nv, nb, 1234 + nb, coherent=False)
        pal = synth.make_palette(nb, 1234)
        ctx.mesh_upload_soa(7, m.pos, m.weights, m.indices, m.normal, m.tangent)
        d_pal = ctx.to_device(pal)
        ref = oracle.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal, m.tangent, threads=8)
        for want in (("pos", "normal", "tangent"), ("pos",), ("normal", "tangent"), ("pos", "tangent")):
            bufs = ctx.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64])
            for b, w in zip(bufs, (3, 3, 4)):
                b.upload(np.full(nv * w + 16, np.nan, np.float32))
            for dyn in (2, 1):
                ctx.set_option("lbs.dyn", dyn)
                for rep in range(3):
                    ctx.lbs_skin_device(7, d_pal.ptr, nb, 1, bufs[0].ptr if "pos" in want else 0, bufs[1].ptr if "normal" in want else 0, bufs[2].ptr if "tangent" in want else 0)
                ctx.sync()
                for k, b, w in (("pos", bufs[0], 3), ("normal", bufs[1], 3), ("tangent", bufs[2], 4)):
                    got = b.download(np.float32, nv * w + 16)
                    if k in want:
                        ok = np.array_equal(got[:nv * w].view(np.uint32), np.ascontiguousarray(ref[k]).reshape(-1).view(np.uint32)) and np.isnan(got[nv * w:]).all()
                    else:
                        ok = np.isnan(got).all()
                    if not ok:
                        bad.append((nv, nb, want, dyn, k))
            for b in bufs:
                b.free()
        d_pal.free()
        ctx.mesh_free(7)
    ctx.set_option("lbs.dyn", 1)
print(json.dumps({"lbs_skin_dma_bit_exact": not bad, "failures": bad[:20]}))
