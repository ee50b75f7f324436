#!/usr/bin/env python3
"""Round 5: one character's frame with the skinning inside the pose launch (fyx_animator_set_skin_output) against update + lbs_skin_device.
C2 (50 k vertices / 64 bones / 1 clip, AnimationPlayer) and C5 (100 k vertices, 4-clip machine).  One JSON line per (config, form).

    python tools/exp/r05_frame_skin.py [frames]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fyrox_amd                      # noqa: E402
from fyrox_amd import anim as A       # noqa: E402
from fyrox_amd import synth           # noqa: E402
import anim_cases as cases            # noqa: E402


def run(ctx, name, sc, mesh, frames):
    nb = sc.rig.n_nodes
    p = cases.build_product(ctx, sc, 1)
    base = p.base_id
    A.create_bone_list(ctx, base + 50, base, list(range(nb)))
    pals = (ctx.malloc(nb * 64), ctx.malloc(nb * 64))
    p.set_palette_output(base + 50, pals[0].ptr)
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    nv = mesh.n_verts
    outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
    update = p.update_machine if sc.machine is not None else p.update_animations

    def timed(fn, n=frames, warm=60):
        for k in range(warm):
            fn(k)
        ctx.sync()
        ctx.timer_begin()
        for k in range(n):
            fn(k)
        return ctx.timer_end() / n * 1e3

    def sep(k):
        update(sc.dt)
        ctx.lbs_skin_device(base + 60, pals[0].ptr, nb, 1, outs[0].ptr, outs[1].ptr, outs[2].ptr)

    def sep_pipe(k):
        dp = pals[k & 1]
        p.set_palette_output(base + 50, dp.ptr)
        update(sc.dt)
        ctx.lbs_skin_device(base + 60, dp.ptr, nb, 1, outs[0].ptr, outs[1].ptr, outs[2].ptr)

    def fused(k):
        update(sc.dt)

    def fused_pipe(k):
        p.set_palette_output(base + 50, pals[k & 1].ptr)
        update(sc.dt)

    rec = {"workload": name}
    ctx.set_option("lbs.streams", 1)
    rec["pose_only_us"] = timed(lambda k: update(sc.dt))
    rec["skin_only_us"] = timed(lambda k: ctx.lbs_skin_device(base + 60, pals[0].ptr, nb, 1, outs[0].ptr, outs[1].ptr, outs[2].ptr))
    rec["separate_us"] = timed(sep)
    ctx.set_option("anim.overlap", 1)
    rec["separate_pipelined_us"] = timed(sep_pipe)
    ctx.set_option("anim.overlap", 0)
    p.set_palette_output(base + 50, pals[0].ptr)
    p.set_skin_output(base + 50, base + 60, outs[0].ptr, outs[1].ptr, outs[2].ptr)
    for units in (0, 1, 2, 4):
        ctx.set_option("anim.frame_skin_units", units)
        rec[f"fused_units{units}_us"] = timed(fused)
        ctx.set_option("anim.overlap", 1)
        rec[f"fused_units{units}_pipelined_us"] = timed(fused_pipe)
        ctx.set_option("anim.overlap", 0)
        p.set_palette_output(base + 50, pals[0].ptr)
    ctx.set_option("anim.frame_skin_units", 0)
    ctx.set_option("anim.frame_skin", 0)
    rec["fallback_launches_us"] = timed(fused)
    ctx.set_option("anim.frame_skin", 1)
    # the same bits?
    ctx.sync()
    a = [b.download(np.uint32, nv * w) for b, w in zip(outs, (3, 3, 4))]
    ctx.lbs_skin_device(base + 60, pals[0].ptr, nb, 1, outs[0].ptr, outs[1].ptr, outs[2].ptr)
    ctx.sync()
    b = [x.download(np.uint32, nv * w) for x, w in zip(outs, (3, 3, 4))]
    rec["bit_identical_to_lbs_skin"] = all(np.array_equal(x, y) for x, y in zip(a, b))
    p.set_skin_output(base + 50, base + 60)
    p.free()
    for x in pals + outs:
        x.free()
    ctx.mesh_free(base + 60)
    print(json.dumps(rec), flush=True)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    with fyrox_amd.Context(0) as ctx:
        rig2 = synth.make_rig(64, synth.SEED_BASE + 2)
        td, tgt = synth.make_clip(64, synth.SEED_BASE + 2, 0)
        c2 = cases.Scenario("c2", rig2, [td], [cases.AnimSpec(0, tgt)], None, n_frames=20)
        run(ctx, "C2", c2, synth.make_mesh(50_000, 64, synth.SEED_BASE + 2), frames)
        run(ctx, "C5", cases.c5_blend_tree(n_bones=64), synth.make_mesh(100_000, 64, synth.SEED_BASE + 5), frames)
        run(ctx, "C2 again", c2, synth.make_mesh(50_000, 64, synth.SEED_BASE + 2), frames)


if __name__ == "__main__":
    main()
