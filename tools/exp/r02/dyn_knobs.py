#!/usr/bin/env python3
"""Experiment: lone-launch time and per-wave timeline of lbs_skin_dyn under its knobs (lbs.dyn_knobs), C4 workload.
    python tools/exp/dyn_knobs.py "256:0,256:8,1024:0x100" > gpurun_out/knobs.jsonl"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fyrox_amd
from fyrox_amd import synth

specs = [(int(a), int(b, 0)) for a, b in (x.split(":") for x in sys.argv[1].split(","))]
extra = [kv.split("=") for kv in sys.argv[2:]]
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
for k, v in extra:
    ctx.set_option(k, int(v))
NV, NB, SETS = 1_000_000, 256, 8
mesh = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
pal = ctx.to_device(synth.make_palette(NB, synth.SEED_BASE + 4))
outs = []
for s in range(SETS):
    ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))


def launch(i):
    s = i % SETS
    ctx.lbs_skin_device(s, pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)


def run(steps):
    ctx.timer_begin()
    for i in range(steps):
        launch(i)
    return ctx.timer_end() * 1e3 / steps


def snap():
    ctx.sync()
    return [outs[0][k].download(np.uint32, NV * w) for k, w in ((0, 3), (1, 3), (2, 4))]


ctx.set_option("lbs.dyn", 0)
launch(0)
ref = snap()
ctx.set_option("lbs.dyn", 1)
times = {}
for rnd in range(3):
    for blk, kn in specs:
        ctx.set_option("lbs.dyn_block", blk)
        ctx.set_option("lbs.dyn_knobs", kn)
        run(20)
        times.setdefault((blk, kn), []).append(run(300))
for blk, kn in specs:
    ctx.set_option("lbs.dyn_block", blk)
    ctx.set_option("lbs.dyn_knobs", kn)
    for b in outs[0]:
        b.upload(np.zeros(b.nbytes // 4, np.uint32))
    launch(0)
    ok = all(np.array_equal(a, b) for a, b in zip(ref, snap()))
    ctx.set_option("lbs.probe", 1)
    n_waves = 4096
    rows = []
    for i in range(8):
        launch(i); launch(i + 1)
        ctx.sync()
        buf = np.zeros((n_waves, 4), np.uint64)
        ctx._check(ctx._l.fyx_debug_read_probe(ctx._h, buf.ctypes.data, n_waves))
        t = buf.astype(np.int64)
        us = (t - t[:, 0].min()) / 100.0
        rows.append([us[:, 0].max(), np.median(us[:, 1]), us[:, 1].max(), np.percentile(us[:, 3], 10), np.median(us[:, 3]),
                     np.percentile(us[:, 3], 90), us[:, 3].max()])
    ctx.set_option("lbs.probe", 0)
    m = np.median(np.array(rows), axis=0)
    ts = times[(blk, kn)]
    print(json.dumps({"block": blk, "knobs": hex(kn), "bit_identical": ok, "lone_us": float(np.median(ts)), "lone_min": float(min(ts)),
                      "entry_last": m[0], "staged_p50": m[1], "staged_max": m[2], "done_p10": m[3], "done_p50": m[4],
                      "done_p90": m[5], "done_max": m[6]}), flush=True)
