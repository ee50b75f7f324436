#!/usr/bin/env python3
"""Experiment: host-side cost of the calls of one C3 crowd frame (1000 instances): the planner alone, the whole update
call, the skinning launch call -- wall clock per call while the GPU runs behind."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
seed = synth.SEED_BASE + 3
rig = synth.make_rig(64, seed)
A.create_rig(ctx, 1, rig)
an = A.Animator(ctx, 1, 1, rig, N)
for c in range(4):
    td, tgt = synth.make_clip(64, seed, clip=c)
    A.upload_tracks_data(ctx, 10 + c, td)
    an.add_animation(10 + c, tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
an.set_machine(synth.make_c5_machine())
for i in range(N):
    for c in range(4):
        an.set_time_position(c, (i * 0.37 + c * 0.11) % 1.0, instance=i)
A.create_bone_list(ctx, 2, 1, list(range(64)))
mesh = synth.make_mesh(10_000, 64, seed)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
nv = 10_000 * N
d_pal = ctx.malloc(N * 64 * 64)
outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
an.set_palette_output(2, d_pal.ptr)
dt = 1 / 60
plan = ctx._l.fyx_animator_plan
upd = ctx._l.fyx_absm_update
skin = ctx._l.fyx_lbs_skin_device
n = ctypes.c_uint32()
res = {}
def t(name, f, reps=200):
    for _ in range(20): f()
    ctx.sync()
    best = 1e9
    for r in range(5):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        el = (time.perf_counter() - t0) / reps
        ctx.sync()
        best = min(best, el)
    res[name] = best * 1e6
cdt = ctypes.c_float(dt)
t("plan_only_us", lambda: plan(ctx._h, an.id, 1, cdt, None, None, None, None, 0, ctypes.byref(n)))
t("update_call_us", lambda: upd(ctx._h, an.id, cdt))
args = (ctx._h, ctypes.c_uint64(3), ctypes.c_void_p(d_pal.ptr), ctypes.c_uint32(64), ctypes.c_uint32(N), ctypes.c_void_p(outs[0].ptr), ctypes.c_void_p(outs[1].ptr), ctypes.c_void_p(outs[2].ptr))
t("skin_call_us (GPU-bound when queued deep)", lambda: skin(*args), reps=50)
def frame():
    upd(ctx._h, an.id, cdt); skin(*args)
t("frame_calls_us", frame, reps=100)
import json; print(json.dumps(res))
