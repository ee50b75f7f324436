#!/usr/bin/env python3
"""GPU box, FYX_LIB_PATH=tools/exp/libs/libfyrox_hip_r05stamp.so (tools/exp/r05_stamps_build.sh): where the one-launch frame that skins
spends its time -- per class of workgroup (sampler / update / skinning), ns after the launch's first stamp, medians over frames.
OPTS=key=value,... sets library options; CFG=c2|c5."""
import ctypes
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                    # noqa: E402
import anim_cases as cases            # noqa: E402
import fyrox_amd                      # noqa: E402
from fyrox_amd import anim as A, synth, _native     # noqa: E402

raw = ctypes.CDLL(_native.LIB_PATH)
ctx = fyrox_amd.Context(0)
for k, v in (kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv):
    ctx.set_option(k, int(v))
names = ["entry", "requests_issued", "samplers_reported", "fold", "local_in_lds", "walk", "palette", "done"]
for name in os.environ.get("CFG", "c2,c5").split(","):
    if name == "c2":
        rig2 = synth.make_rig(64, synth.SEED_BASE + 2)
        td, tgt = synth.make_clip(64, synth.SEED_BASE + 2, 0)
        sc = cases.Scenario("c2", rig2, [td], [cases.AnimSpec(0, tgt)], None, n_frames=20)
        mesh = synth.make_mesh(50_000, 64, synth.SEED_BASE + 2)
    else:
        sc = cases.c5_blend_tree(n_bones=64)
        mesh = synth.make_mesh(100_000, 64, synth.SEED_BASE + 5)
    p = cases.build_product(ctx, sc, 1)
    nb = sc.rig.n_nodes
    base = p.base_id
    A.create_bone_list(ctx, base + 50, base, list(range(nb)))
    d_pal = ctx.malloc(nb * 64)
    p.set_palette_output(base + 50, d_pal.ptr)
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    nv = mesh.n_verts
    outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
    if os.environ.get("SKIN", "1") == "1":
        p.set_skin_output(base + 50, base + 60, outs[0].ptr, outs[1].ptr, outs[2].ptr)
    update = p.update_machine if sc.machine is not None else p.update_animations
    n_samp = ((nb * 16 + 255) // 256) * len(sc.animations)
    per_class = {"sampler": [], "update": [], "skin_first": [], "skin_median": [], "skin_last": []}
    n_skin = 0
    for f in range(60):
        buf = np.zeros(1024 * 8, np.uint64)
        # clear the stamps: a workgroup that did not run this frame must not show last frame's
        update(sc.dt)
        ctx.sync()
        assert raw.fyx_exp_frame_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        st = buf.reshape(1024, 8).astype(np.int64)
        t0 = st[:n_samp, 0].min()
        live = st[:, 0] >= t0
        rel = (st - t0) * 10      # ns
        if f < 20:
            continue
        per_class["sampler"].append(np.median(rel[:n_samp], axis=0))
        per_class["update"].append(rel[n_samp])
        sk = rel[n_samp + 1:][live[n_samp + 1:]]
        n_skin = len(sk)
        if n_skin:
            per_class["skin_first"].append(sk.min(axis=0))
            per_class["skin_median"].append(np.median(sk, axis=0))
            per_class["skin_last"].append(sk.max(axis=0))
    out = {"workload": name, "opts": os.environ.get("OPTS", ""), "sampler_workgroups": n_samp, "skin_workgroups": n_skin}
    for k, rows in per_class.items():
        if rows:
            med = np.median(np.array(rows), axis=0).astype(int).tolist()
            out[k] = dict(zip(names, med)) if k != "sampler" else {"entry": med[0], "counter_added": med[7]}
    print(json.dumps(out), flush=True)
    p.set_skin_output(base + 50, base + 60)
    p.free()
    for b in outs:
        b.free()
    d_pal.free()
    ctx.mesh_free(base + 60)
ctx.close()
