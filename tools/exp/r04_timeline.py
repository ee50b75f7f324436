#!/usr/bin/env python3
"""Round 4 experiment: WHERE the C3 frame's kernels run relative to each other (option debug.timeline: every pose_sample /
pose_update / skinning launch carries its own events -- no tracer, whose interception made the pipelined frame 2.3 x slower),
and what the HOST spends per call.  One JSON line per configuration:
  period_us        median distance between consecutive skinning starts
  skin_us / sample_us / update_us   median kernel durations inside the loop
  sample_start_after_prev_skin_start_us   > 0: the pose update of frame n + 1 starts this long after frame n's skinning started
  skin_start_after_update_stop_us         the gap between a frame's update kernel and its skinning
  host: upd_call_us / skin_call_us / setpal_us    wall clock inside each C-ABI call (includes waiting when the GPU is behind)
Env: VERTS (default 10000; 16 = a mesh so small that the host and the pose kernels are all that is left)."""
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
VERTS = int(os.environ.get("VERTS", "10000"))
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
mesh = synth.make_mesh(VERTS, 64, seed)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
nv = VERTS * N
pals = [ctx.malloc(N * 64 * 64) for _ in range(2)]
outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
cdt = ctypes.c_float(1 / 60)
upd, skin, setpal = ctx._l.fyx_absm_update, ctx._l.fyx_lbs_skin_device, ctx._l.fyx_animator_set_palette_output
SK = [(ctx._h, ctypes.c_uint64(3), ctypes.c_void_p(p.ptr), ctypes.c_uint32(64), ctypes.c_uint32(N), ctypes.c_void_p(outs[0].ptr),
       ctypes.c_void_p(outs[1].ptr), ctypes.c_void_p(outs[2].ptr)) for p in pals]
n_made = 0


def make_animator():
    global n_made
    an = A.Animator(ctx, 100 + n_made, 1, rig, N)
    n_made += 1
    for c in range(4):
        an.add_animation(10 + c, tds[c], time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
    an.set_machine(synth.make_c5_machine())
    for i in range(N):
        for c in range(4):
            an.set_time_position(c, (i * 0.37 + c * 0.11) % 1.0, instance=i)
    return an


def run(an, frames, pipelined, acc=None):
    aid = ctypes.c_uint64(an.id)
    pc = time.perf_counter
    for k in range(frames):
        b = k & 1 if pipelined else 0
        t0 = pc()
        setpal(ctx._h, aid, ctypes.c_uint64(2), ctypes.c_void_p(pals[b].ptr))
        t1 = pc()
        upd(ctx._h, aid, cdt)
        t2 = pc()
        skin(*SK[b])
        t3 = pc()
        if acc is not None:
            acc[0] += t1 - t0
            acc[1] += t2 - t1
            acc[2] += t3 - t2


def measure(name, opts, pipelined):
    for k, v in opts.items():
        ctx.set_option(k, v)
    an = make_animator()
    run(an, 60, pipelined)
    ctx.sync()
    frames = 300
    acc = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    ctx.timer_begin()
    run(an, frames, pipelined, acc)
    host = time.perf_counter() - t0
    frame_us = ctx.timer_end() / frames * 1e3
    ctx.set_option("debug.timeline", 1)
    run(an, 40, pipelined)
    kinds, a, b = ctx.timeline()
    ctx.set_option("debug.timeline", 0)
    sk = [(x, y) for k_, x, y in zip(kinds, a, b) if k_ == 0]
    sa = [(x, y) for k_, x, y in zip(kinds, a, b) if k_ == 1]
    up = [(x, y) for k_, x, y in zip(kinds, a, b) if k_ == 2]
    cp = [(x, y) for k_, x, y in zip(kinds, a, b) if k_ == 3]
    n = min(len(sk), len(sa), len(up))
    med = lambda v: round(float(np.median(v)), 2) if len(v) else None
    rec = {"config": name, "verts": VERTS, "frame_us": round(frame_us, 2), "host_loop_us_per_frame": round(host / frames * 1e6, 2),
           "host": {"setpal_us": round(acc[0] / frames * 1e6, 2), "upd_call_us": round(acc[1] / frames * 1e6, 2), "skin_call_us": round(acc[2] / frames * 1e6, 2)},
           "timeline_frames": n,
           "period_us": med([sk[i + 1][0] - sk[i][0] for i in range(5, n - 1)]),
           "skin_us": med([y - x for x, y in sk[5:n]]), "sample_us": med([y - x for x, y in sa[5:n]]), "update_us": med([y - x for x, y in up[5:n]]),
           "update_start_after_sample_stop_us": med([up[i][0] - sa[i][1] for i in range(5, n)]),
           "skin_start_after_update_stop_us": med([sk[i][0] - up[i][1] for i in range(5, n)]),
           "sample_start_after_prev_skin_start_us": med([sa[i][0] - sk[i - 1][0] for i in range(5, n)]),
           "sample_start_after_prev_skin_stop_us": med([sa[i][0] - sk[i - 1][1] for i in range(5, n)]),
           "skin_start_after_prev_skin_stop_us": med([sk[i][0] - sk[i - 1][1] for i in range(5, n)]),
           "copy_us": med([y - x for x, y in cp[5:n]]) if len(cp) >= n else None,
           "copy_start_after_prev_update_stop_us": med([cp[i][0] - up[i - 1][1] for i in range(5, n)]) if len(cp) >= n else None,
           "sample_start_after_copy_stop_us": med([sa[i][0] - cp[i][1] for i in range(5, n)]) if len(cp) >= n else None,
           "frames_10_to_13": [{"sample": [round(sa[i][0] - sk[10][0], 1), round(sa[i][1] - sk[10][0], 1)], "update": [round(up[i][0] - sk[10][0], 1), round(up[i][1] - sk[10][0], 1)],
                                "skin": [round(sk[i][0] - sk[10][0], 1), round(sk[i][1] - sk[10][0], 1)]} for i in range(10, min(14, n))],
           "opts": opts}
    print(json.dumps(rec), flush=True)
    an.free()
    ctx.set_option("anim.overlap", 0)
    ctx.set_option("lbs.streams", 1)


base = {"lbs.streams": 1, "anim.overlap": 0}
pipe = {"lbs.streams": 1, "anim.overlap": 1}
measure("one chain on one stream", base, False)
measure("frames alternate between two streams", pipe, True)
ctx.set_option("lbs.exact", 0)
measure("one chain on one stream, fused skinning", base, False)
measure("frames alternate between two streams, fused skinning", pipe, True)
ctx.set_option("lbs.exact", 1)
ctx.close()
