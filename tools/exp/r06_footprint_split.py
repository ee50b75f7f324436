#!/usr/bin/env python3
"""Round 6: WHAT of the rotation's footprint the lone C4 launch pays for.  The counters (profiles/r06_footprint_counters.json) say the
knee is memory-side read latency, not translation; this separates inputs from outputs and "how many sets" from "which sets": the launch
rotates over n_in input sets (meshes) and n_out output sets independently.  One JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth


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
        outs.append(ctx.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64]))
    res = {}

    def case(name, ins, os_):
        st = {"k": 0}

        def rot():
            k = st["k"]
            st["k"] += 1
            o = outs[os_[k % len(os_)]]
            ctx.lbs_skin_device(1 + ins[k % len(ins)], pal.ptr, nb, 1, o[0].ptr, o[1].ptr, o[2].ptr)
        res.setdefault(name, []).append(round(kernel_us(ctx, rot, 800), 2))


    for rep in range(2):
        case("in1_out1", [0], [0])
        case("in8_out1", list(range(8)), [0])
        case("in1_out8", [0], list(range(8)))
        case("in16_out1", list(range(16)), [0])
        case("in1_out16", [0], list(range(16)))
        case("in8_out8", list(range(8)), list(range(8)))
        case("in6_out6_first", list(range(6)), list(range(6)))
        case("in6_out6_last", list(range(10, 16)), list(range(10, 16)))
        case("in6_out6_every_other", list(range(0, 12, 2)), list(range(0, 12, 2)))
        case("in16_out16", list(range(16)), list(range(16)))
        case("in3_out8", list(range(3)), list(range(8)))
        case("in8_out3", list(range(8)), list(range(3)))
        case("in12_out4", list(range(12)), list(range(4)))
        case("in4_out12", list(range(4)), list(range(12)))
    print(json.dumps({"what": "lbs_skin_dyn kernel_us; rotation over n_in input sets (60 MB each) and n_out output sets (40 MB each)", "kernel_us": res}))
