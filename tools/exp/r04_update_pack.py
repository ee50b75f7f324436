#!/usr/bin/env python3
"""Round 4 experiment: the crowd's update launch with 1 / 2 / 4 instances per workgroup (option anim.update_pack), C3 frame,
interleaved in one process: overlap {0, 1} x pack {0, 2, 4}, five rounds.  frame_us by HIP events over 300 frames after 60 warm-up
frames; EXACT env 0 = fused skinning.  First the palettes of one frame under every pack value, compared bit for bit."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth
N = int(os.environ.get("N", "1000"))
PACKS = [int(x) for x in os.environ.get("PACKS", "0,2,4").split(",")]
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
ctx.set_option("lbs.exact", int(os.environ.get("EXACT", "1")))
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
cdt = ctypes.c_float(1 / 60)
upd, skin, setpal = ctx._l.fyx_absm_update, ctx._l.fyx_lbs_skin_device, ctx._l.fyx_animator_set_palette_output
SK = [(ctx._h, ctypes.c_uint64(3), ctypes.c_void_p(p.ptr), ctypes.c_uint32(64), ctypes.c_uint32(N), ctypes.c_void_p(outs[0].ptr), ctypes.c_void_p(outs[1].ptr),
       ctypes.c_void_p(outs[2].ptr)) for p in pals]


def make_animator(aid):
    an = A.Animator(ctx, aid, 1, rig, N)
    for c in range(4):
        an.add_animation(10 + c, tds[c], time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
    an.set_machine(synth.make_c5_machine())
    for i in range(N):
        for c in range(4):
            an.set_time_position(c, (i * 0.37 + c * 0.11) % 1.0, instance=i)
    return an


# parity of the forms: three animators in the same state, three frames each, palettes bit for bit
ref = None
for k, pack in enumerate(PACKS):
    ctx.set_option("anim.update_pack", pack)
    an = make_animator(200 + k)
    for _ in range(3):
        setpal(ctx._h, ctypes.c_uint64(an.id), ctypes.c_uint64(2), ctypes.c_void_p(pals[0].ptr))
        upd(ctx._h, ctypes.c_uint64(an.id), cdt)
    ctx.sync()
    got = pals[0].download(np.uint32, N * 64 * 16)
    if ref is None:
        ref = got.copy()
    print(json.dumps({"pack": pack, "palettes_bit_equal_to_pack0": bool(np.array_equal(ref, got)), "nonzero": int(np.count_nonzero(got))}), flush=True)

an = make_animator(100)
aid = ctypes.c_uint64(an.id)
frame_no = 0


def run(frames, ring):
    global frame_no
    for _ in range(frames):
        b = frame_no % ring
        frame_no += 1
        setpal(ctx._h, aid, ctypes.c_uint64(2), ctypes.c_void_p(pals[b].ptr))
        upd(ctx._h, aid, cdt)
        skin(*SK[b])


keys = [(m, p) for m in (0, 1) for p in PACKS]
res = {k: [] for k in keys}
for rnd in range(5):
    for (mode, pack) in keys:
        ctx.set_option("anim.update_pack", pack)
        ctx.set_option("anim.overlap", mode)
        ring = (1, 2)[mode]
        run(60, ring)
        ctx.sync()
        ctx.timer_begin()
        run(300, ring)
        res[(mode, pack)].append(round(ctx.timer_end() / 300 * 1e3, 2))
        ctx.set_option("anim.overlap", 0)
for (m, p) in keys:
    print(json.dumps({"anim.overlap": m, "anim.update_pack": p, "frame_us_rounds": res[(m, p)], "frame_us_median": float(np.median(res[(m, p)]))}), flush=True)
ctx.close()
