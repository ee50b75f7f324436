#!/usr/bin/env python3
"""Lone-launch time of the single-instance skinning kernel under its work-distribution options, on the C4 workload
(1 M vertices / 256 bones, 8 rotating 100 MB buffer sets), ONE launch stream so that the HIP-event average per launch
is the kernel's duration as rocprofv3 --kernel-trace reports it.  Every variant's output is first compared bit for bit
with the plain static kernel's (which the test-suite pins against the oracle).

    python tools/tune_dyn.py [--quick] > gpurun_out/tune_dyn.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fyrox_amd
from fyrox_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--verts", type=int, default=1_000_000)
ap.add_argument("--bones", type=int, default=256)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--streams", default="1,2")
ap.add_argument("--only", default="", help="comma-separated variant names to keep")
ap.add_argument("--tag", default="")
args = ap.parse_args()

ctx = fyrox_amd.Context(0)
mesh = synth.make_mesh(args.verts, args.bones, synth.SEED_BASE + 4)
pal = synth.make_palette(args.bones, synth.SEED_BASE + 4)
d_pal = ctx.to_device(pal)
nv = mesh.n_verts
outs = []
for s in range(args.sets):
    ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs.append((ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)))

BASE = {"lbs.block": 512, "lbs.blocks_per_cu": 4, "lbs.prefetch": 1, "lbs.exact": 1, "lbs.nt": 1, "lbs.split": 0,
        "lbs.dyn": 0, "lbs.dyn_bpc": 0, "lbs.dyn_block": 256, "lbs.asym": 0, "lbs.young_prio": 0}


def configure(opts, streams):
    o = dict(BASE)
    o.update(opts)
    for k, v in o.items():
        ctx.set_option(k, v)
    ctx.set_option("lbs.streams", streams)


def run(steps):
    ctx.timer_begin()
    for i in range(steps):
        s = i % args.sets
        ctx.lbs_skin_device(s, d_pal.ptr, args.bones, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
    return ctx.timer_end() * 1e3 / steps


def snapshot():
    ctx.sync()
    return [outs[0][0].download(np.uint32, nv * 3), outs[0][1].download(np.uint32, nv * 3),
            outs[0][2].download(np.uint32, nv * 4)]


def clear():
    z = np.zeros(nv * 4, np.uint32)
    for b in outs[0]:
        b.upload(z[: (b.nbytes // 4)])


variants = [("static p1 bpc4", {}), ("static p1 bpc2", {"lbs.blocks_per_cu": 2}),
            ("static p3 bpc2", {"lbs.blocks_per_cu": 2, "lbs.prefetch": 3}),
            ("static p3 bpc4", {"lbs.prefetch": 3})]
for a in (36, 40, 44):
    variants.append((f"static p1 bpc2 asym{a}", {"lbs.blocks_per_cu": 2, "lbs.asym": a}))
variants.append(("static p3 bpc2 asym40", {"lbs.blocks_per_cu": 2, "lbs.asym": 40, "lbs.prefetch": 3}))
for pr in (1, 3):
    variants.append((f"static p1 bpc2 prio{pr}", {"lbs.blocks_per_cu": 2, "lbs.young_prio": pr}))
for blk in (1024, 512, 256):
    variants.append((f"dyn b{blk}", {"lbs.dyn_block": blk, "lbs.dyn": 1}))
    variants.append((f"dyn b{blk} exact0", {"lbs.dyn_block": blk, "lbs.dyn": 1, "lbs.exact": 0}))
variants.append(("dyn b256 bpc2", {"lbs.dyn_block": 256, "lbs.dyn": 1, "lbs.dyn_bpc": 2}))
variants.append(("dyn b512 bpc1", {"lbs.dyn_block": 512, "lbs.dyn": 1, "lbs.dyn_bpc": 1}))
variants.append(("static p1 bpc4 exact0", {"lbs.exact": 0}))

if args.only:
    keep = set(args.only.split(","))
    variants = [v for v in variants if v[0] in keep]

# ---- correctness: every variant against the plain static kernel, bit for bit ---------------------------------------
configure({}, 1)
clear()
run(1)
ref = snapshot()
bad = []
for name, o in variants:
    if o.get("lbs.exact", 1) == 0:
        continue
    configure(o, 1)
    clear()
    run(1)
    got = snapshot()
    if not all(np.array_equal(a, b) for a, b in zip(ref, got)):
        bad.append(name)
# the drawn kernel again over a few launches in a row on two streams
configure({"lbs.dyn": 1, "lbs.dyn_block": 1024}, 2)
for rep in range(3):
    clear()
    for i in range(args.sets * 2 + 1):
        s = i % args.sets
        ctx.lbs_skin_device(s, d_pal.ptr, args.bones, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
    got = snapshot()
    if not all(np.array_equal(a, b) for a, b in zip(ref, got)):
        bad.append(f"dyn repeated launches, rep {rep}")
print("# mismatching variants:", bad, file=sys.stderr)

# ---- timing ---------------------------------------------------------------------------------------------------------
results = {}
for streams in [int(x) for x in args.streams.split(",")]:
    for r in range(args.rounds):
        for name, o in variants:
            configure(o, streams)
            run(20)
            results.setdefault((name, streams), []).append(run(args.steps))
rows = []
for (name, streams), ts in results.items():
    us = float(np.median(ts))
    rows.append({"variant": name, "streams": streams, "us_median": us, "us_min": float(min(ts)),
                 "frac_of_8TBps": 100.0 * nv / us / 1e3 / 8000.0})
rows.sort(key=lambda r: (r["streams"], r["us_median"]))
print(json.dumps({"tag": args.tag, "lib": os.environ.get("FYX_LIB_PATH", ""), "verts": nv, "bones": args.bones, "mismatching": bad, "rows": rows}, indent=1))
for r in rows:
    print("#", r, file=sys.stderr)
