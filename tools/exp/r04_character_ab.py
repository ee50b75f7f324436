#!/usr/bin/env python3
"""Round 4 experiment driver: one character's frame (C2: player, 50 k vertices; C5: 4-clip machine, 100 k vertices; 64 bones)
under sets of options, interleaved in one process, five rounds.  SETS="k=v,k=v;k=v".  frame_us / pose_us (no skinning) by HIP
events over 400 frames after 100 warm-up frames, everything on one stream."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth
import anim_cases as cases
SETS = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in st.split(",") if kv) for st in os.environ.get("SETS", "").split(";")]
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
base = {k: ctx.get_option(k) for st in SETS for k in st}
out = {}
for name, sc, nverts in (("c2", cases.player_only(n_bones=64, seed=synth.SEED_BASE + 2), 50_000), ("c5", cases.c5_blend_tree(), 100_000)):
    nb = sc.rig.n_nodes
    p = cases.build_product(ctx, sc, 1)
    A.create_bone_list(ctx, p.base_id + 50, p.base_id, list(range(nb)))
    d_pal = ctx.malloc(nb * 64)
    p.set_palette_output(p.base_id + 50, d_pal.ptr)
    mesh = synth.make_mesh(nverts, nb, synth.SEED_BASE + 9)
    ctx.mesh_upload_soa(p.base_id + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    d_pos, d_nrm, d_tan = ctx.malloc(nverts * 12 + 64), ctx.malloc(nverts * 12 + 64), ctx.malloc(nverts * 16 + 64)
    update = p.update_machine if sc.machine is not None else p.update_animations

    def run(n, skin):
        for _ in range(n):
            update(sc.dt)
            if skin:
                ctx.lbs_skin_device(p.base_id + 60, d_pal.ptr, nb, 1, d_pos.ptr, d_nrm.ptr, d_tan.ptr)

    res = {(i, k): [] for i in range(len(SETS)) for k in ("frame", "pose")}
    for rnd in range(5):
        for i, st in enumerate(SETS):
            for k, v in base.items():
                ctx.set_option(k, v)
            for k, v in st.items():
                ctx.set_option(k, v)
            for kind, skin in (("frame", True), ("pose", False)):
                run(100, skin)
                ctx.sync()
                ctx.timer_begin()
                run(400, skin)
                res[(i, kind)].append(round(ctx.timer_end() / 400 * 1e3, 2))
    for i, st in enumerate(SETS):
        print(json.dumps({"workload": name, "options": st, "frame_us": res[(i, "frame")], "frame_us_median": float(np.median(res[(i, "frame")])),
                          "pose_us": res[(i, "pose")], "pose_us_median": float(np.median(res[(i, "pose")]))}), flush=True)
    p.free()
ctx.close()
