"""GPU box, with FYX_LIB_PATH=tools/exp/libs/libfyrox_hip_updstamp.so (a build of the library whose pose_update kernel writes
eight wall_clock64 stamps -- 100 MHz -- over the first local matrix of instance 0): where one character's update spends its time.
Stamps: 0 entry, 1 top-of-kernel requests issued + program known, 2 fold done, 3 local matrices in LDS (barrier passed),
4 hierarchy walk done, 5 matrices copied out, 6 palette stores issued, 7 all stores acknowledged."""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import anim_cases as cases
import fyrox_amd
from fyrox_amd import anim as A, synth

ctx = fyrox_amd.Context(0)
out = {}
for name, sc in (("c5", cases.c5_blend_tree(n_bones=64)), ("layered", cases.layered()), ("transitions", cases.transitions())):
    p = cases.build_product(ctx, sc, 1)
    nb = sc.rig.n_nodes
    A.create_bone_list(ctx, p.base_id + 50, p.base_id, list(range(nb)))
    d_pal = ctx.malloc(nb * 64)
    p.set_palette_output(p.base_id + 50, d_pal.ptr)
    rows, mhz, lv = [], [], []
    for f in range(60):
        p.update_machine(sc.dt)
        ctx.sync()
        allw = p.read(A.READ_LOCAL_MATRIX).reshape(-1).view(np.uint64)[:17].astype(np.int64)
        raw = allw[:9]
        if f >= 20:
            lv.append(np.diff(allw[9:17]))
        st = raw[:8]
        if f >= 20:
            rows.append(np.diff(st) * 10)     # ns
            mhz.append(float(raw[8]) / max(1.0, float(st[7] - st[0]) * 10) * 1e3)
    rows = np.array(rows)
    out[name] = {"levels": int(sc.rig.n_levels) if hasattr(sc.rig, "n_levels") else None, "nodes": nb,
                 "median_ns_between_stamps": np.median(rows, axis=0).astype(int).tolist(), "total_ns": int(np.median(rows.sum(axis=1))),
                 "cycles_between_the_starts_of_levels_0_to_7": np.median(np.array(lv), axis=0).astype(int).tolist(),
                 "shader_clock_mhz": round(float(np.median(mhz)), 1)}
print(json.dumps(out))
ctx.close()
