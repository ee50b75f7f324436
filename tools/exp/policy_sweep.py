#!/usr/bin/env python3
"""Cache-policy sweep of lbs_skin_dyn's streams (experiment build, FYX_EXP_POLICY): lone-launch and two-stream time per
(load policy, store policy), each first checked bit for bit against the static kernel."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fyrox_amd
from fyrox_amd import synth

NV, NB, SETS = 1_000_000, 256, 8
ctx = fyrox_amd.Context(0)
mesh = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
d_pal = ctx.to_device(synth.make_palette(NB, synth.SEED_BASE + 4))
outs = []
for s in range(SETS):
    ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))


def run(steps):
    ctx.timer_begin()
    for i in range(steps):
        s = i % SETS
        ctx.lbs_skin_device(s, d_pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
    return ctx.timer_end() * 1e3 / steps


def snapshot():
    ctx.sync()
    return [outs[0][0].download(np.uint32, NV * 3), outs[0][1].download(np.uint32, NV * 3), outs[0][2].download(np.uint32, NV * 4)]


def clear():
    z = np.zeros(NV * 4, np.uint32)
    for b in outs[0]:
        b.upload(z[: b.nbytes // 4])


names = ["plain", "nt", "sc1", "sc0sc1", "sc1nt"]
ctx.set_option("lbs.streams", 1)
ctx.set_option("lbs.blocks_per_cu", 2)
run(1); ref = snapshot()
ctx.set_option("lbs.dyn", 1)
rows = []
for pol in range(0, 26):
    ctx.set_option("lbs.policy", pol)
    ctx.set_option("lbs.streams", 1)
    clear(); run(1)
    ok = all(np.array_equal(a, b) for a, b in zip(ref, snapshot()))
    t1 = []
    t2 = []
    for r in range(3):
        ctx.set_option("lbs.streams", 1); run(20); t1.append(run(300))
        ctx.set_option("lbs.streams", 2); run(20); t2.append(run(300))
    label = "default" if pol == 0 else f"ld={names[(pol - 1) // 5]} st={names[(pol - 1) % 5]}"
    rows.append({"policy": pol, "label": label, "bit_identical": ok, "lone_us": float(np.median(t1)), "two_stream_us": float(np.median(t2))})
    print("# %2d %-28s ok=%s lone %6.2f  2-stream %6.2f" % (pol, label, ok, rows[-1]["lone_us"], rows[-1]["two_stream_us"]), file=sys.stderr)
ctx.set_option("lbs.dyn", 0); ctx.set_option("lbs.policy", 0)
for bpc in (2, 4):
    ctx.set_option("lbs.blocks_per_cu", bpc)
    ctx.set_option("lbs.streams", 1); run(20); a = run(300)
    ctx.set_option("lbs.streams", 2); run(20); b = run(300)
    rows.append({"policy": -bpc, "label": f"static bpc{bpc}", "bit_identical": True, "lone_us": a, "two_stream_us": b})
    print("# static bpc%d lone %6.2f 2-stream %6.2f" % (bpc, a, b), file=sys.stderr)
print(json.dumps(rows))
