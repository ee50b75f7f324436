#!/usr/bin/env python3
"""BASELINE config C3 end to end on one MI355X: a crowd of N instances of one character
(64-bone rig, 4 clips, the C5 blend-tree machine), per frame:
    fyx_absm_update (host control plane + pose_sample + pose_update kernels)
 -> fyx_animator_palette (palette_gather)
 -> fyx_lbs_skin_device (instanced skinning: N x verts_per_instance vertices, per-instance palette in LDS)
all resident in HBM.  Prints one JSON line with per-stage times (HIP events / host clock)."""
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
ap.add_argument("--instances", type=int, default=1000)
ap.add_argument("--verts", type=int, default=10_000)
ap.add_argument("--bones", type=int, default=64)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--opt", action="append", default=["lbs.streams=1"],
                help="kernel option key=value; the frame is a dependent chain (pose -> palette -> skin), so the "
                     "default keeps every launch on the context stream instead of forking to the worker streams")
ap.add_argument("--palette-output", action="store_true",
                help="let the update kernel write the palettes itself (fyx_animator_set_palette_output) instead of a separate gather launch")
ap.add_argument("--root-motion", action="store_true",
                help="RootMotionSettings on every clip (root = node 0) + AnimationPose::root_motion tracking")
args = ap.parse_args()

ctx = fyrox_amd.Context(0)
for kv in args.opt:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
seed = synth.SEED_BASE + 3
rig = synth.make_rig(args.bones, seed)
A.create_rig(ctx, 1, rig)
an = A.Animator(ctx, 1, 1, rig, args.instances)
for c in range(4):
    td, tgt = synth.make_clip(args.bones, seed, clip=c)
    A.upload_tracks_data(ctx, 10 + c, td)
    an.add_animation(10 + c, tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
an.set_machine(synth.make_c5_machine())
if args.root_motion:
    for c in range(4):
        an.set_root_motion_settings(c, 0)
# desynchronise the crowd: every instance starts at its own phase and stands at its own place
for i in range(args.instances):
    for c in range(4):
        an.set_time_position(c, (i * 0.37 + c * 0.11) % 1.0, instance=i)
place = np.zeros((args.instances, 10), np.float32)
place[:, 0] = (np.arange(args.instances) % 40) * 2.0
place[:, 2] = (np.arange(args.instances) // 40) * 2.0
place[:, 6] = 1.0
place[:, 7:10] = 1.0
an.set_local_trs(0, place)
A.create_bone_list(ctx, 2, 1, list(range(args.bones)))
mesh = synth.make_mesh(args.verts, args.bones, seed)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
nv = args.verts * args.instances
d_pal = ctx.malloc(args.instances * args.bones * 64)
d_pos, d_nrm, d_tan = ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)
dt = 1.0 / 60.0


if args.palette_output:
    an.set_palette_output(2, d_pal.ptr)


def frame(skin=True):
    an.update_machine(dt)
    if not args.palette_output:
        an.palette(2, d_pal.ptr)
    if skin:
        ctx.lbs_skin_device(3, d_pal.ptr, args.bones, args.instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)


for _ in range(args.warmup):
    frame()
ctx.sync()
# whole frame
t0 = time.perf_counter()
ctx.timer_begin()
for _ in range(args.frames):
    frame()
gpu_ms = ctx.timer_end()
wall = time.perf_counter() - t0
# pose part only (control plane + sample + update + palette)
ctx.sync()
t0 = time.perf_counter()
ctx.timer_begin()
for _ in range(args.frames):
    frame(skin=False)
pose_gpu_ms = ctx.timer_end()
pose_wall = time.perf_counter() - t0
# host control plane alone
t0 = time.perf_counter()
for _ in range(args.frames):
    an.plan(1, dt)
plan_wall = time.perf_counter() - t0
# skinning alone
ctx.timer_begin()
for _ in range(args.frames):
    ctx.lbs_skin_device(3, d_pal.ptr, args.bones, args.instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
skin_ms = ctx.timer_end()

unique_bytes = args.verts * 60 + args.instances * args.bones * 64 + nv * 40
out = {
    "workload": f"C3 crowd: {args.instances} instances x {args.verts} verts / {args.bones} bones, 4-clip blend-tree machine per instance",
    "frame_ms_gpu": gpu_ms / args.frames, "frame_ms_wall": wall * 1e3 / args.frames,
    "pose_ms_gpu": pose_gpu_ms / args.frames, "pose_ms_wall": pose_wall * 1e3 / args.frames,
    "host_control_plane_ms": plan_wall * 1e3 / args.frames,
    "skin_ms_gpu": skin_ms / args.frames,
    "skinned_vertices_per_s": nv / (skin_ms / args.frames * 1e-3),
    "skin_unique_hbm_GBps": unique_bytes / (skin_ms / args.frames * 1e-3) / 1e9,
    "skin_frac_of_8TBps": unique_bytes / (skin_ms / args.frames * 1e-3) / 1e9 / 8000.0,
    "crowd_frames_per_s": args.frames / wall,
    "bones_posed_per_s": args.instances * args.bones * args.frames / pose_wall,
}
print(json.dumps(out), flush=True)
ctx.close()
