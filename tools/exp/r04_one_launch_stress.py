#!/usr/bin/env python3
"""Round 4: the one-launch frame against the two-launch frame, bit for bit, over many frames (FRAMES env, default 20000): two
animators of the same scenario in the same state, one updated with anim.one_launch = 1, the other with 0, a skinning launch of
100 k vertices behind each (memory traffic beside the next frame's kernels); palettes compared after every frame.  A record the
update half read before the sampler half's store had landed would show as a difference (the previous frame's value)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth
import anim_cases as cases
FRAMES = int(os.environ.get("FRAMES", "20000"))
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
res = {}
for name, mk in (("c5", lambda: cases.c5_blend_tree()), ("c2", lambda: cases.player_only(n_bones=64, seed=synth.SEED_BASE + 2)), ("transitions", cases.transitions)):
    sc = mk()
    nb = sc.rig.n_nodes
    ps, pals = [], []
    for k in range(2):
        p = cases.build_product(ctx, sc, 1)
        A.create_bone_list(ctx, p.base_id + 50, p.base_id, list(range(nb)))
        d = ctx.malloc(nb * 64)
        p.set_palette_output(p.base_id + 50, d.ptr)
        ps.append(p); pals.append(d)
    mesh = synth.make_mesh(100_000, nb, synth.SEED_BASE + 9)
    ctx.mesh_upload_soa(ps[0].base_id + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs = (ctx.malloc(100_000 * 12 + 64), ctx.malloc(100_000 * 12 + 64), ctx.malloc(100_000 * 16 + 64))
    bad = 0
    first_bad = None
    for f in range(FRAMES):
        for idx, par in sc.script.get(f % 64, []):
            for p in ps:
                p.set_parameter(idx, par)
        for k, one in ((0, 1), (1, 0)):
            ctx.set_option("anim.one_launch", one)
            (ps[k].update_machine if sc.machine is not None else ps[k].update_animations)(sc.dt)
            ctx.lbs_skin_device(ps[0].base_id + 60, pals[k].ptr, nb, 1, outs[0].ptr, outs[1].ptr, outs[2].ptr)
        a = pals[0].download(np.uint32, nb * 16)
        b = pals[1].download(np.uint32, nb * 16)
        if not np.array_equal(a, b):
            bad += 1
            if first_bad is None:
                first_bad = f
    res[name] = {"frames": FRAMES, "frames_that_differ": bad, "first": first_bad}
    print(json.dumps({name: res[name]}), flush=True)
    for p in ps:
        p.free()
ctx.set_option("anim.one_launch", 1)
ctx.close()
