#!/usr/bin/env python3
"""Experiment: lbs_skin_crowd (every lane stores its own vertex) against lbs_skin_crowd_flat (position / normal leave
through an LDS strip as line-aligned 16-byte pieces), option lbs.crowd_stage, on crowds of 1000 instances x VERTS
vertices / 64 bones: kernel duration (lbs.timing) and launch period, exact and fused, interleaved A/B in one process;
bit-identity of the two forms is asserted on every configuration first."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth

inst, bones = int(os.environ.get("INST", "1000")), 64
vert_list = [int(v) for v in os.environ.get("VERTS", "10000").split(",")]
ipbs = [int(v) for v in os.environ.get("IPB", "0").split(",")]
reps = int(os.environ.get("REPS", "60"))
STAGES = [int(v) for v in os.environ.get("STAGES", "0").split(",")]   # variant numbers of tools/exp/r03_crowd_forms.patch (0 = the product kernel)
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); ctx.set_option(k, int(v))
for verts in vert_list:
    seed = synth.SEED_BASE + 3
    mesh = synth.make_mesh(verts, bones, seed)
    pal = synth.make_palette(bones, seed, n_instances=inst)
    ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    nv = verts * inst
    d_pal = ctx.to_device(pal)
    outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))

    def launch():
        ctx.lbs_skin_device(3, d_pal.ptr, bones, inst, outs[0].ptr, outs[1].ptr, outs[2].ptr)

    # parity of the two forms (whole buffers, bit for bit), exact mode
    ctx.set_option("lbs.exact", 1)
    got = {}
    for stage in [st for st in STAGES if st not in (3, 7, 8, 9, 10, 11)]:
        ctx.set_option("lbs.crowd_stage", stage)
        for o, w in zip(outs, (12, 12, 16)):
            o.upload(np.full(nv * w // 4, 0xFFFFFFFF, dtype=np.uint32))     # a byte the form does not write stays visible
        launch(); ctx.sync()
        got[stage] = [o.download(np.uint32, nv * w // 4) for o, w in zip(outs, (12, 12, 16))]
    same = all(np.array_equal(a, b) for st in STAGES[1:] if st not in (3, 7, 8, 9, 10, 11) for a, b in zip(got[STAGES[0]], got[st]))
    print(json.dumps({"verts": verts, "forms_bit_identical": bool(same)}), flush=True)
    del got
    for ipb in ipbs:
        ctx.set_option("lbs.crowd_ipb", ipb)
        for exact in (1, 0):
            ctx.set_option("lbs.exact", exact)
            res = {st: ([], []) for st in STAGES}
            for r in range(3):
                for stage in STAGES:
                    ctx.set_option("lbs.crowd_stage", stage)
                    for _ in range(5): launch()
                    ctx.sync(); ctx.timer_begin()
                    for _ in range(reps): launch()
                    res[stage][0].append(ctx.timer_end() / reps * 1e3)
                    ctx.set_option("lbs.timing", 1); ctx.kernel_time()
                    for _ in range(reps): launch()
                    us, n = ctx.kernel_time(); ctx.set_option("lbs.timing", 0)
                    res[stage][1].append(us / n)
            unique = verts * 60 + inst * bones * 64 + nv * 40
            for stage in STAGES:
                k = float(np.median(res[stage][1]))
                print(json.dumps({"verts": verts, "ipb": ipb, "exact": exact, "stage": stage,
                                  "period_us": round(float(np.median(res[stage][0])), 2), "kernel_us": round(k, 2),
                                  "frac": round(unique / (k * 1e-6) / 8e12, 4)}), flush=True)
    for o in outs: o.free()
    d_pal.free()
ctx.close()
