#!/usr/bin/env python3
"""Round 4 experiment: the C3 crowd FRAME (1000 instances x 10 k vertices / 64 bones, 4-clip machine per instance) under the
stream / upload / kernel-form options -- one chain on one stream, and pipelined (anim.overlap: frame n + 1's pose update
beside frame n's skinning, two palette buffers) with the pose stream prioritised or CU-masked.
Per configuration: frame period by HIP events over the whole loop, host time per frame (the loop's wall clock without the
final sync), and a bit-for-bit check of the pipelined frames against the serial ones.  One JSON line per configuration."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth

N = int(os.environ.get("N_INST", "1000"))
FRAMES = int(os.environ.get("FRAMES", "300"))
ctx = fyrox_amd.Context(0)
seed = synth.SEED_BASE + 3
rig = synth.make_rig(64, seed)
A.create_rig(ctx, 1, rig)
tds = []
for c in range(4):
    td, tgt = synth.make_clip(64, seed, clip=c)
    A.upload_tracks_data(ctx, 10 + c, td)
    tds.append(tgt)
A.create_bone_list(ctx, 2, 1, list(range(64)))
mesh = synth.make_mesh(10_000, 64, seed)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
nv = 10_000 * N
pals = [ctx.malloc(N * 64 * 64) for _ in range(2)]
outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
dt = 1 / 60
cdt = ctypes.c_float(dt)
upd = ctx._l.fyx_absm_update
skin = ctx._l.fyx_lbs_skin_device
setpal = ctx._l.fyx_animator_set_palette_output


def make_animator(aid, root_motion=False):
    an = A.Animator(ctx, aid, 1, rig, N)
    for c in range(4):
        an.add_animation(10 + c, tds[c], time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
    an.set_machine(synth.make_c5_machine())
    if root_motion:
        for c in range(4):
            an.set_root_motion_settings(c, 0)
        an.track_root_motion(True)
    for i in range(N):
        for c in range(4):
            an.set_time_position(c, (i * 0.37 + c * 0.11) % 1.0, instance=i)
    return an


def skin_args(pal):
    return (ctx._h, ctypes.c_uint64(3), ctypes.c_void_p(pal.ptr), ctypes.c_uint32(64), ctypes.c_uint32(N), ctypes.c_void_p(outs[0].ptr),
            ctypes.c_void_p(outs[1].ptr), ctypes.c_void_p(outs[2].ptr))


SK = [skin_args(p) for p in pals]


def run(an, frames, pipelined):
    aid = ctypes.c_uint64(an.id)
    if pipelined:
        for k in range(frames):
            setpal(ctx._h, aid, ctypes.c_uint64(2), ctypes.c_void_p(pals[k & 1].ptr))
            upd(ctx._h, aid, cdt)
            skin(*SK[k & 1])
    else:
        setpal(ctx._h, aid, ctypes.c_uint64(2), ctypes.c_void_p(pals[0].ptr))
        for k in range(frames):
            upd(ctx._h, aid, cdt)
            skin(*SK[0])


def measure(name, opts, pipelined, root_motion=False, check=None):
    for k, v in opts.items():
        ctx.set_option(k, v)
    an = make_animator(100 + measure.n, root_motion)
    measure.n += 1
    run(an, 40, pipelined)
    ctx.sync()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.timer_begin()
        run(an, FRAMES, pipelined)
        host = time.perf_counter() - t0
        gpu_ms = ctx.timer_end()
        r = (gpu_ms / FRAMES * 1e3, host / FRAMES * 1e6)
        best = r if best is None or r[0] < best[0] else best
    total = 40 + 3 * FRAMES
    ctx.sync()
    # what the last frame produced (palette buffer of frame total - 1; the outputs)
    last = pals[(total - 1) & 1] if pipelined else pals[0]
    pal = last.download(np.uint32, N * 64 * 16)
    pos = outs[0].download(np.uint32, 3 * 10_000 * 8)      # the first eight instances' positions
    rec = {"config": name, "frame_us": round(best[0], 2), "host_us_per_frame": round(best[1], 2), "opts": opts, "pipelined": pipelined}
    if check is not None:
        rec["equals_serial"] = bool(np.array_equal(pal, check[0]) and np.array_equal(pos, check[1]))
    print(json.dumps(rec), flush=True)
    an.free()
    ctx.set_option("anim.overlap", 0)
    ctx.set_option("lbs.streams", 1)
    ctx.set_option("lbs.crowd_lean", 0)
    return pal, pos


measure.n = 0
base = {"lbs.streams": 1, "anim.overlap": 0}
which = os.environ.get("WHICH", "all")
ref = measure("serial, upload stream (r03)", {**base, "anim.ctrl_upload": 0, "anim.update_lean": 0}, False)
if which in ("all", "serial"):
    measure("serial, upload stream, lean update", {**base, "anim.ctrl_upload": 0, "anim.update_lean": 1}, False, check=ref)
    measure("serial, copy in stream, lean update", {**base, "anim.ctrl_upload": 1, "anim.update_lean": 1}, False, check=ref)
    measure("serial, copy kernel, lean update", {**base, "anim.ctrl_upload": 2, "anim.update_lean": 1}, False, check=ref)
pipe = {"lbs.streams": 2, "anim.overlap": 1}
if which in ("all", "pipe"):
    for prio in (0, 1):
        for lean_upd in (0, 1):
            for lean_crowd in (0, 1):
                for cm in (0, 1, 2):
                    if cm != 1 and not (prio == 1 and lean_upd == 1):
                        continue
                    measure(f"pipelined prio={prio} update_lean={lean_upd} crowd_lean={lean_crowd} ctrl={cm}",
                            {**pipe, "streams.priority": prio, "streams.pose_cus": 0, "anim.update_lean": lean_upd, "lbs.crowd_lean": lean_crowd,
                             "anim.ctrl_upload": cm}, True, check=ref)
    for cus in (16, 32, 64):
        for lean_crowd in (0, 1):
            measure(f"pipelined pose_cus={cus} crowd_lean={lean_crowd}", {**pipe, "streams.pose_cus": cus, "anim.update_lean": 1, "lbs.crowd_lean": lean_crowd,
                                                                          "anim.ctrl_upload": 1}, True, check=ref)
    ctx.set_option("streams.pose_cus", 0)
    ctx.set_option("streams.priority", 1)
if which == "trace":      # one pipelined configuration, for a kernel trace (tools/exp/r04_overlap.py reads it)
    measure("pipelined prio=1 update_lean=1 crowd_lean=%s ctrl=1 (trace)" % os.environ.get("CROWD_LEAN", "0"),
            {**pipe, "streams.priority": int(os.environ.get("PRIO", "1")), "streams.pose_cus": int(os.environ.get("POSE_CUS", "0")), "anim.update_lean": 1,
             "lbs.crowd_lean": int(os.environ.get("CROWD_LEAN", "0")), "anim.ctrl_upload": 1}, True, check=ref)
if which in ("all", "rm"):
    rref = measure("root motion, serial", {**base, "anim.ctrl_upload": 0, "anim.update_lean": 1}, False, root_motion=True)
    measure("root motion, pipelined prio", {**pipe, "streams.priority": 1, "anim.update_lean": 1, "anim.ctrl_upload": 1}, True, root_motion=True, check=rref)
ctx.close()
