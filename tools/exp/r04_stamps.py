#!/usr/bin/env python3
"""GPU box, with FYX_LIB_PATH=tools/exp/libs/libfyrox_hip_r04stamp.so (tools/exp/r04_stamps_build.sh): where one character's pose
kernel spends its time.  ns between the stamps of the update workgroup's thread 0 (see the build script), medians over 40 frames."""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import anim_cases as cases
import fyrox_amd
from fyrox_amd import anim as A, synth

ctx = fyrox_amd.Context(0)
for k, v in (kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv):
    ctx.set_option(k, int(v))
names = ["top", "wait_for_sampler", "fold", "local_matrices", "walk", "matrices_out", "palette", "stores_acknowledged"]
for name, sc in (("c2", cases.player_only(n_bones=64, seed=synth.SEED_BASE + 2)), ("c5", cases.c5_blend_tree(n_bones=64)), ("transitions", cases.transitions())):
    p = cases.build_product(ctx, sc, 1)
    nb = sc.rig.n_nodes
    A.create_bone_list(ctx, p.base_id + 50, p.base_id, list(range(nb)))
    d_pal = ctx.malloc(nb * 64)
    p.set_palette_output(p.base_id + 50, d_pal.ptr)
    update = p.update_machine if sc.machine is not None else p.update_animations
    rows, srows = [], []
    for f in range(60):
        update(sc.dt)
        ctx.sync()
        allw = p.read(A.READ_LOCAL_MATRIX).reshape(-1).view(np.uint64)[:17].astype(np.int64)
        st = allw[:9]
        if f >= 20:
            rows.append(np.diff(st) * 10)     # ns
            srows.append((allw[9:17] - st[0]) * 10)      # the sampler's stamps, ns after the update workgroup's entry
    rows = np.array(rows)
    med = np.median(rows, axis=0).astype(int).tolist()
    print(json.dumps({"workload": name, "nodes": nb, "ns": dict(zip(names, med)), "total_ns": int(np.median(rows.sum(axis=1))),
                      "sampler_block0_thread0_ns_after_update_entry": dict(zip(["entry", "desc_time_tick", "hint", "span_value", "store_issued", "release_fence", "barrier", "counter_added"],
                                                                               np.median(np.array(srows), axis=0).astype(int).tolist()))}), flush=True)
    p.free()
ctx.close()
