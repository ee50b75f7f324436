#!/usr/bin/env python3
"""60 000 one-launch frames that skin (fyx_animator_set_skin_output) beside a chip-filling skinning launch of another mesh, against an
animator in the same state run as separate launches: palettes and vertices bit for bit every 2000 frames.  C5's machine, 20 k vertices.
One JSON line.  python tools/exp/r05_frame_skin_stress.py [frames]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fyrox_amd
from fyrox_amd import anim as A, synth
import anim_cases as cases

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
with fyrox_amd.Context(0) as ctx:
    sc = cases.c5_blend_tree(euler_every=10 ** 6)
    nb, nv = sc.rig.n_nodes, 20_000
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 27)
    big = synth.make_mesh(1_000_000, nb, synth.SEED_BASE + 28)
    ctx.mesh_upload_soa(1, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    ctx.mesh_upload_soa(2, big.pos, big.weights, big.indices, big.normal, big.tangent)
    big_out = (ctx.malloc(12_000_064), ctx.malloc(12_000_064), ctx.malloc(16_000_064))
    big_pal = ctx.to_device(synth.make_palette(nb, synth.SEED_BASE + 28))
    ps, pals, outs = [], [], []
    for k in range(2):
        p = cases.build_product(ctx, sc, 1)
        A.create_bone_list(ctx, p.base_id + 50, p.base_id, list(range(nb)))
        d = ctx.malloc(nb * 64)
        p.set_palette_output(p.base_id + 50, d.ptr)
        ps.append(p); pals.append(d)
        outs.append((ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)))
    ps[0].set_skin_output(ps[0].base_id + 50, 1, outs[0][0].ptr, outs[0][1].ptr, outs[0][2].ptr)
    t0 = time.time()
    checks, bad = 0, 0
    for f in range(frames):
        ctx.lbs_skin_device(2, big_pal.ptr, nb, 1, big_out[0].ptr, big_out[1].ptr, big_out[2].ptr)
        ctx.set_option("anim.one_launch", 1)
        ps[0].update_machine(sc.dt)
        ctx.set_option("anim.one_launch", 0)
        ps[1].update_machine(sc.dt)
        ctx.lbs_skin_device(1, pals[1].ptr, nb, 1, outs[1][0].ptr, outs[1][1].ptr, outs[1][2].ptr)
        if f % 2000 == 1999:
            checks += 1
            same = np.array_equal(pals[0].download(np.uint32, nb * 16), pals[1].download(np.uint32, nb * 16))
            for a, b, w in zip(outs[0], outs[1], (3, 3, 4)):
                same &= np.array_equal(a.download(np.uint32, nv * w), b.download(np.uint32, nv * w))
            bad += 0 if same else 1
    ctx.sync()
    print(json.dumps({"frames": frames, "checks": checks, "mismatches": bad, "seconds": round(time.time() - t0, 1),
                      "what": "one-launch frames that skin vs separate launches, a 1 M-vertex skinning launch in flight beside every frame; palettes + three vertex streams bit for bit"}))
