#!/usr/bin/env python3
"""Round 6: the lone C4 launch (lbs_skin_dyn, 1 M vertices / 256 bones, per-dispatch events) against the number of rotating 100 MB
sets, for each value of an option (default lbs.dyn_map = 0 / 1), the forms interleaved in ONE process on ONE set of allocations.
    python tools/exp/r06_sets_sweep.py [option] [values,comma]
One JSON line: {option, values, kernel_us_by_sets_and_value: {k: {value: [us, us]}}}."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth

OPT = sys.argv[1] if len(sys.argv) > 1 else "lbs.dyn_map"
VALUES = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]


def kernel_us(ctx, launch, n, warm=40):
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
    nv, nb, MAXS = 1_000_000, 256, 16
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 4)
    pal = ctx.to_device(synth.make_palette(nb, synth.SEED_BASE + 4))
    outs = []
    for m in range(MAXS):
        ctx.mesh_upload_soa(1 + m, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        outs.append((ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)))
    res = {}
    for sets in (1, 2, 4, 6, 8, 12, 16, 8, 6, 1):
        for val in VALUES:
            ctx.set_option(OPT, val)
            st = {"k": 0}

            def rot():
                k = st["k"] % sets
                st["k"] += 1
                o = outs[k]
                ctx.lbs_skin_device(1 + k, pal.ptr, nb, 1, o[0].ptr, o[1].ptr, o[2].ptr)
            res.setdefault(str(sets), {}).setdefault(str(val), []).append(round(kernel_us(ctx, rot, 800), 2))
    print(json.dumps({"option": OPT, "values": VALUES, "kernel_us_by_sets_and_value": res}))
