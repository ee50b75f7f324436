#!/usr/bin/env python3
"""GPU box, FYX_LIB_PATH=tools/exp/libs/libfyrox_hip_r06sstamp.so (tools/exp/r06_scene_stamps_build.sh): where the scene's sampler
(pose_sample_scene_kernel, 256 characters x 4 clips x 64 nodes = 4096 workgroups) spends its time -- per workgroup: when it started
(ns after the launch's first stamp), when its job parameters had arrived, when thread 0's curve was sampled, when its stores were
issued.  One JSON line: percentiles over the workgroups of a frame, medians over frames."""
import ctypes, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import anim as A, synth, _native

raw = ctypes.CDLL(_native.LIB_PATH)
ctx = fyrox_amd.Context(0)
K, N, NB = 256, 1, 64
animators = []
for k in range(K):
    seed = synth.SEED_BASE + 100 + k
    rig = synth.make_rig(NB, seed)
    A.create_rig(ctx, 1000 + k, rig)
    an = A.Animator(ctx, 2000 + k, 1000 + k, rig, N)
    for c in range(4):
        td, tgt = synth.make_clip(NB, seed, clip=c)
        A.upload_tracks_data(ctx, 10_000 + 4 * k + c, td)
        an.add_animation(10_000 + 4 * k + c, tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
    an.set_machine(synth.make_c5_machine())
    for c in range(4):
        an.set_time_position(c, (c * 0.11 + k * 0.05) % 1.0, instance=0)
    A.create_bone_list(ctx, 3000 + k, 1000 + k, list(range(NB)))
    d_pal = ctx.malloc(N * NB * 64)
    an.set_palette_output(3000 + k, d_pal.ptr)
    animators.append(an)
n_wg = K * 4 * ((NB * 16 + 255) // 256)
rows = []
per_frame = []
for f in range(50):
    raw.fyx_exp_scene_stamps_clear()
    A.scene_update(ctx, animators, 1.0 / 60.0)
    ctx.sync()
    buf = np.zeros(8192 * 4, np.uint64)
    assert raw.fyx_exp_scene_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(8192, 4)[:n_wg].astype(np.int64)
    paths = np.zeros(4, np.uint32)
    assert raw.fyx_exp_scene_paths(paths.ctypes.data_as(ctypes.c_void_p)) == 0
    per_frame.append([int(x) for x in paths] + [float((st[:, 3].max() - st[:, 0].min()) * 10.0)])
    if f < 10:
        continue
    t0 = st[:, 0].min()
    rel = (st - t0) * 10.0      # ns
    start, params, sampled, done = rel[:, 0], rel[:, 1] - rel[:, 0], rel[:, 2] - rel[:, 1], rel[:, 3] - rel[:, 2]
    pct = lambda x: [float(np.percentile(x, q)) for q in (5, 50, 95, 100)]
    rows.append({"start": pct(start), "entry_to_params": pct(params), "params_to_sampled": pct(sampled), "sampled_to_stores": pct(done),
                 "lifetime": pct(rel[:, 3] - rel[:, 0]), "last_done": float(rel[:, 3].max()),
                 "started_by_us": [int((start <= t).sum()) for t in (1000, 2000, 4000, 6000, 8000, 10000, 12000, 16000, 20000)]})
med = {k: np.median(np.asarray([r[k] for r in rows], float), axis=0).round(0).tolist() for k in rows[0]}
print(json.dumps({"what": "pose_sample_scene_kernel, 256 x 1 x 64 nodes x 4 clips; ns; percentiles 5 / 50 / 95 / 100 over the 4096 workgroups, medians over 40 frames",
                  "n_workgroups": n_wg, **med, "started_by_us_at": [1, 2, 4, 6, 8, 10, 12, 16, 20],
                  "per_frame_curve_samples_by_exit_inside_next_prev_general_then_kernel_ns": per_frame}))
