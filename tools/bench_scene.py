#!/usr/bin/env python3
"""A heterogeneous scene on one MI355X: K distinct characters (each its own rig, clips, state machine and mesh --
one AnimationPlayer + AnimationBlendingStateMachine + Mesh per character in the reference's scene graph,
`fyrox-impl/src/scene/animation/absm.rs:311-326`), N instances of each.  Per frame and per character:
    fyx_scene_update (all characters at once; `one_by_one` = fyx_absm_update per character) -> palettes (written by the
    update kernel) -> fyx_lbs_skin_batch (`one_by_one`: fyx_lbs_skin_device per character)
Many small dependent chains: launch-bound rather than bandwidth-bound.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fyrox_amd
from fyrox_amd import anim as A
from fyrox_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--characters", type=int, default=64)
ap.add_argument("--instances", type=int, default=4)
ap.add_argument("--verts", type=int, default=20_000)
ap.add_argument("--bones", type=int, default=64)
ap.add_argument("--frames", type=int, default=100)
ap.add_argument("--warmup", type=int, default=60)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--shared-clips", action="store_true", help="diagnostic: every character plays the SAME four tracks data (their span records stay in cache)")
ap.add_argument("--batched-only", action="store_true", help="skip the one-by-one legs (short runs under a profiler)")
args = ap.parse_args()

ctx = fyrox_amd.Context(0)
for kv in args.opt:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
K, N = args.characters, args.instances
chars = []
for k in range(K):
    seed = synth.SEED_BASE + 100 + k
    rig = synth.make_rig(args.bones, seed)
    rid, aid, bid, mid = 1000 + k, 2000 + k, 3000 + k, 4000 + k
    A.create_rig(ctx, rid, rig)
    an = A.Animator(ctx, aid, rid, rig, N)
    for c in range(4):
        td, tgt = synth.make_clip(args.bones, synth.SEED_BASE + 100 if args.shared_clips else seed, clip=c)
        tid = 10_000 + c if args.shared_clips else 10_000 + 4 * k + c
        if not (args.shared_clips and k > 0):
            A.upload_tracks_data(ctx, tid, td)
        an.add_animation(tid, tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
    an.set_machine(synth.make_c5_machine())
    for i in range(N):
        for c in range(4):
            an.set_time_position(c, (i * 0.37 + c * 0.11 + k * 0.05) % 1.0, instance=i)
    A.create_bone_list(ctx, bid, rid, list(range(args.bones)))
    mesh = synth.make_mesh(args.verts, args.bones, seed)
    ctx.mesh_upload_soa(mid, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    nv = args.verts * N
    d_pal = ctx.malloc(N * args.bones * 64)
    outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
    an.set_palette_output(bid, d_pal.ptr)
    chars.append((an, mid, d_pal, outs))
dt = 1.0 / 60.0


animators = [c[0] for c in chars]
# the id array is built once: a Python list comprehension over 256 objects per frame costs more than the library's whole host path
import ctypes
_ids = np.asarray([a.id for a in animators], np.uint64)
_scene_update = ctx._l.fyx_scene_update
_ids_p, _n_ids, _dt = _ids.ctypes.data_as(ctypes.c_void_p), len(_ids), ctypes.c_float(1.0 / 60.0)


from fyrox_amd._native import SkinJob
skin_jobs = (SkinJob * K)(*[SkinJob(mid, d_pal.ptr, args.bones, N, o[0].ptr, o[1].ptr, o[2].ptr) for _, mid, d_pal, o in chars])


def frame(skin=True, pose=True, batched=True):
    if batched:
        if pose:
            ctx._check(_scene_update(ctx._h, _ids_p, _n_ids, _dt))   # one launch per stage for the whole scene
        if skin:
            ctx.lbs_skin_batch(skin_jobs)            # one launch for every mesh of the scene
        return
    for an, mid, d_pal, o in chars:                  # one by one: what the batched calls replace
        if pose:
            an.update_machine(dt)
        if skin:
            ctx.lbs_skin_device(mid, d_pal.ptr, args.bones, N, o[0].ptr, o[1].ptr, o[2].ptr)


def timed(**kw):
    """Median of three passes of `frames` frames (a single 10 - 15 ms pass right after set-up is not always at speed)."""
    res = []
    for _ in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(args.frames):
            frame(**kw)
        gpu = ctx.timer_end()
        res.append((gpu / args.frames, (time.perf_counter() - t0) * 1e3 / args.frames))
    res.sort(key=lambda r: r[1])
    return res[1]


for _ in range(args.warmup):
    frame()
f_gpu, f_wall = timed()
p_gpu, p_wall = timed(skin=False)
s_gpu, s_wall = timed(pose=False)
if args.batched_only:
    f1_gpu = f1_wall = p1_gpu = p1_wall = s1_gpu = s1_wall = None
else:
    f1_gpu, f1_wall = timed(batched=False)
    p1_gpu, p1_wall = timed(skin=False, batched=False)
    s1_gpu, s1_wall = timed(pose=False, batched=False)
_scene_plan = ctx._l.fyx_scene_plan
t0 = time.perf_counter()
for _ in range(args.frames):
    ctx._check(_scene_plan(ctx._h, _ids_p, _n_ids, _dt))      # the host half of fyx_scene_update alone (planner threads included)
plan = (time.perf_counter() - t0) * 1e3 / args.frames
total_verts = K * N * args.verts
print(json.dumps({
    "workload": f"{K} distinct characters x {N} instances x {args.verts} verts / {args.bones} bones, 4-clip blend-tree machine each",
    "options": {k: ctx.get_option(k) for k in ("lbs.streams", "anim.threads")},
    "frame_ms_gpu": f_gpu, "frame_ms_wall": f_wall, "pose_ms_gpu": p_gpu, "pose_ms_wall": p_wall,
    "one_by_one": {"frame_ms_wall": f1_wall, "pose_ms_gpu": p1_gpu, "pose_ms_wall": p1_wall, "skin_ms_gpu": s1_gpu, "skin_ms_wall": s1_wall},
    "skin_algorithmic_GBps": total_verts * 100 / (s_gpu * 1e-3) / 1e9,
    "skin_ms_gpu": s_gpu, "skin_ms_wall": s_wall, "host_control_plane_ms": plan,
    "per_character_us_wall": f_wall * 1e3 / K, "skinned_vertices_per_s": total_verts / (f_wall * 1e-3),
    "scene_frames_per_s": 1e3 / f_wall}), flush=True)
ctx.close()
