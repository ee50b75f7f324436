#!/usr/bin/env python3
"""Round 4 experiment driver: the scene tick (N_CHARS distinct characters x N_INST instances, 64 bones, 4-clip machine each; one
fyx_scene_update + one fyx_lbs_skin_batch per frame).  Prints per-frame GPU time by events (frame / pose only / skinning only) and
the HOST time of the fyx_scene_update call (perf_counter over calls that never wait for the GPU: a sync every 8 frames, outside the
clock).  Under rocprofv3 --kernel-trace the kernel durations come with it."""
import ctypes as ct, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth
from fyrox_amd._native import SkinJob
N_CHARS, N_INST, N_VERTS = int(os.environ.get("N_CHARS", "256")), int(os.environ.get("N_INST", "1")), int(os.environ.get("N_VERTS", "5000"))
FRAMES = int(os.environ.get("FRAMES", "200"))
ctx = fyrox_amd.Context(0)
for k, v in (kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv):
    ctx.set_option(k, int(v))
nb, dt = 64, 1.0 / 60.0
chars = []
for k in range(N_CHARS):
    seed = synth.SEED_BASE + 700 + k
    rig = synth.make_rig(nb, seed)
    rid, aid, bid, mid = (1000 + j * 10_000 + k for j in range(4))
    tid = 200_000 + 4 * k
    A.create_rig(ctx, rid, rig)
    an = A.Animator(ctx, aid, rid, rig, N_INST)
    for c in range(4):
        td, tgt = synth.make_clip(nb, seed, clip=c, euler_every=10 ** 9)
        A.upload_tracks_data(ctx, tid + c, td)
        an.add_animation(tid + c, tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
    an.set_machine(synth.make_c5_machine())
    A.create_bone_list(ctx, bid, rid, list(range(nb)))
    mesh = synth.make_mesh(N_VERTS, nb, seed)
    ctx.mesh_upload_soa(mid, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    nv = N_VERTS * N_INST
    d_pal = ctx.malloc(N_INST * nb * 64)
    outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
    an.set_palette_output(bid, d_pal.ptr)
    chars.append((an, mid, d_pal, outs))
ids = np.asarray([c[0].id for c in chars], np.uint64)
ids_p, cdt = ids.ctypes.data_as(ct.c_void_p), ct.c_float(dt)
jobs = (SkinJob * N_CHARS)(*[SkinJob(mid, d_pal.ptr, nb, N_INST, o[0].ptr, o[1].ptr, o[2].ptr) for _, mid, d_pal, o in chars])
upd, batch = ctx._l.fyx_scene_update, ctx._l.fyx_lbs_skin_batch


def frame(pose=True, skin=True):
    if pose:
        ctx._check(upd(ctx._h, ids_p, N_CHARS, cdt))
    if skin:
        ctx._check(batch(ctx._h, jobs, N_CHARS))


for _ in range(80):
    frame()


def timed(**kw):
    ctx.sync()
    ctx.timer_begin()
    for _ in range(FRAMES):
        frame(**kw)
    return round(ctx.timer_end() / FRAMES * 1e3, 2)


res = {"workload": f"scene {N_CHARS} x {N_INST} x {N_VERTS}", "options": os.environ.get("OPTS", ""), "frame_us": [timed() for _ in range(3)],
       "pose_us": [timed(skin=False) for _ in range(3)], "skin_us": [timed(pose=False) for _ in range(3)]}
host = []
for _ in range(FRAMES // 8):
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        ctx._check(upd(ctx._h, ids_p, N_CHARS, cdt))
    host.append((time.perf_counter() - t0) / 8 * 1e6)
res["host_us_per_scene_update_call"] = round(float(np.median(host)), 1)
host = []
for _ in range(FRAMES // 8):
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        ctx._check(batch(ctx._h, jobs, N_CHARS))
    host.append((time.perf_counter() - t0) / 8 * 1e6)
res["host_us_per_skin_batch_call"] = round(float(np.median(host)), 1)
print(json.dumps(res), flush=True)
ctx.close()
