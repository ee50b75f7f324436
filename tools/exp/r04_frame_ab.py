#!/usr/bin/env python3
"""Round 4 experiment driver: the C3 frame under sets of options, interleaved in one process, five rounds each, one chain
(anim.overlap 0) and alternating frame streams (1).  SETS="k=v,k=v;k=v" (';' between sets, the first is the base line; every
key of any set is reset to the base value between sets).  frame_us by HIP events over 300 frames after 60 warm-up frames;
EXACT env 0 = fused skinning."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth
N = int(os.environ.get("N", "1000"))
SETS = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in st.split(",") if kv) for st in os.environ.get("SETS", "").split(";")]
MODES = [int(x) for x in os.environ.get("MODES", "0,1").split(",")]
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
ctx.set_option("lbs.exact", int(os.environ.get("EXACT", "1")))
base = {k: ctx.get_option(k) for st in SETS for k in st}
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
an = A.Animator(ctx, 100, 1, rig, N)
for c in range(4):
    an.add_animation(10 + c, tds[c], time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
an.set_machine(synth.make_c5_machine())
for i in range(N):
    for c in range(4):
        an.set_time_position(c, (i * 0.37 + c * 0.11) % 1.0, instance=i)
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


keys = [(m, i) for m in MODES for i in range(len(SETS))]
res = {k: [] for k in keys}
for rnd in range(5):
    for (mode, i) in keys:
        for k, v in base.items():
            ctx.set_option(k, v)
        for k, v in SETS[i].items():
            ctx.set_option(k, v)
        ctx.set_option("anim.overlap", mode)
        ring = (1, 2)[mode]
        run(60, ring)
        ctx.sync()
        ctx.timer_begin()
        run(300, ring)
        res[(mode, i)].append(round(ctx.timer_end() / 300 * 1e3, 2))
        ctx.set_option("anim.overlap", 0)
for (m, i) in keys:
    print(json.dumps({"anim.overlap": m, "options": SETS[i], "frame_us_rounds": res[(m, i)], "frame_us_median": float(np.median(res[(m, i)]))}), flush=True)
ctx.close()
