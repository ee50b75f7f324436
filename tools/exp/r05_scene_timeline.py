#!/usr/bin/env python3
"""Round 5: which kernels of a pipelined scene frame run beside which.  Two uses:

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/exp/r05_scene_timeline.py run pipelined|serial [chars verts instances]
    python tools/exp/r05_scene_timeline.py read DIR          # the trace's kernels, frame by frame (last 100 frames)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(mode, n_chars=256, n_verts=5000, n_inst=1, frames=300):
    import numpy as np
    import fyrox_amd
    from fyrox_amd import anim as A, synth
    nb, dt = 64, 1.0 / 60.0
    with fyrox_amd.Context(0) as ctx:
        chars = []
        for k in range(n_chars):
            seed = synth.SEED_BASE + 700 + k
            rig = synth.make_rig(nb, seed)
            rid, aid, bid, mid = (9_000_000 + j * 10_000 + k for j in range(4))
            tid = 9_100_000 + 4 * k
            A.create_rig(ctx, rid, rig)
            an = A.Animator(ctx, aid, rid, rig, n_inst)
            for c in range(4):
                td, tgt = synth.make_clip(nb, seed, clip=c, euler_every=10 ** 9)
                A.upload_tracks_data(ctx, tid + c, td)
                an.add_animation(tid + c, tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
            an.set_machine(synth.make_c5_machine())
            A.create_bone_list(ctx, bid, rid, list(range(nb)))
            mesh = synth.make_mesh(n_verts, nb, seed)
            ctx.mesh_upload_soa(mid, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
            pals = (ctx.malloc(n_inst * nb * 64), ctx.malloc(n_inst * nb * 64))
            outs = (ctx.malloc(n_inst * n_verts * 12 + 64), ctx.malloc(n_inst * n_verts * 12 + 64), ctx.malloc(n_inst * n_verts * 16 + 64))
            an.set_palette_output_pair(bid, pals[0].ptr, pals[1].ptr)
            an.set_skin_output(bid, mid, outs[0].ptr, outs[1].ptr, outs[2].ptr)
            chars.append(an)
        ctx.set_option("anim.overlap", 1 if mode == "pipelined" else 0)
        for _ in range(60):
            A.scene_update(ctx, chars, dt)
        ctx.sync()
        ctx.timer_begin()
        for _ in range(frames):
            A.scene_update(ctx, chars, dt)
        ms = ctx.timer_end() / frames
        print(json.dumps({"mode": mode, "frame_us": ms * 1e3}), flush=True)


def read(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fyx::", ""), r.get("Queue_Id", "")))
    rows.sort()
    names = ("ctrl_copy", "pose_sample_scene", "pose_update", "lbs_skin_batch")
    rows = [r for r in rows if any(n in r[2] for n in names)]
    tail = rows[-400:]
    t0 = tail[0][0]
    for s, e, n, q in tail[:48]:
        print(f"{(s - t0) / 1e3:9.2f} -> {(e - t0) / 1e3:9.2f}  ({(e - s) / 1e3:6.2f} us)  q{q}  {n[:40]}")
    import statistics
    for n in names:
        ds = [(e - s) / 1e3 for s, e, k, q in tail if n in k]
        if ds:
            print(f"{n:20s} n={len(ds):4d} median {statistics.median(ds):6.2f} us  mean {statistics.fmean(ds):6.2f}")
    starts = [s for s, e, k, q in tail if "pose_sample_scene" in k]
    print("frame period (sampler start to sampler start), median us:", statistics.median([(b - a) / 1e3 for a, b in zip(starts, starts[1:])]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], *(int(x) for x in sys.argv[3:]))
    else:
        read(sys.argv[2])
