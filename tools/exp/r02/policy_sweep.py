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


import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--policies", default="0,7,8,6,9")
ap.add_argument("--blocks", default="512,256,1024")
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
names = ["plain", "nt", "sc1", "sc0sc1", "sc1nt"]
ctx.set_option("lbs.streams", 1)
ctx.set_option("lbs.blocks_per_cu", 2)
run(1); ref = snapshot()
cfgs = [("static bpc2", {"lbs.dyn": 0, "lbs.block": 512, "lbs.blocks_per_cu": 2, "lbs.policy": 0}),
        ("static bpc4", {"lbs.dyn": 0, "lbs.block": 512, "lbs.blocks_per_cu": 4, "lbs.policy": 0})]
for blk in [int(x) for x in args.blocks.split(",")]:
    for pol in [int(x) for x in args.policies.split(",")]:
        label = "default(nt,nt)" if pol == 0 else f"ld={names[(pol - 1) // 5]} st={names[(pol - 1) % 5]}"
        cfgs.append((f"dyn b{blk} {label}", {"lbs.dyn": 1, "lbs.dyn_block": blk, "lbs.blocks_per_cu": 2, "lbs.policy": pol}))
res = {}
for name, o in cfgs:
    for k, v in o.items():
        ctx.set_option(k, v)
    ctx.set_option("lbs.streams", 1)
    clear(); run(1)
    res[name] = {"ok": all(np.array_equal(a, b) for a, b in zip(ref, snapshot())), "t1": [], "t2": []}
for r in range(args.rounds):
    for name, o in cfgs:
        for k, v in o.items():
            ctx.set_option(k, v)
        ctx.set_option("lbs.streams", 1); run(20); res[name]["t1"].append(run(300))
        ctx.set_option("lbs.streams", 2); run(20); res[name]["t2"].append(run(300))
rows = []
for name, r in res.items():
    rows.append({"config": name, "bit_identical": r["ok"], "lone_us": float(np.median(r["t1"])), "lone_min": float(min(r["t1"])),
                 "two_stream_us": float(np.median(r["t2"]))})
    print("# %-36s ok=%s lone %6.2f (min %6.2f)  2-stream %6.2f" % (name, r["ok"], rows[-1]["lone_us"], rows[-1]["lone_min"], rows[-1]["two_stream_us"]), file=sys.stderr)
print(json.dumps(rows))
