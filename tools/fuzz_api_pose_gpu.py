#!/usr/bin/env python3
"""API-sequence fuzz of the POSE path on the GPU box: a handful of live animators (random machines) are updated frame by frame -- one by
one or through fyx_scene_update -- against one oracle each, while between the frames calls with wrong arguments are thrown at them: reads
of selectors that do not exist, palette outputs of unknown bone lists, skin outputs of unknown meshes, set_local_trs past the rig or past
the instances, scene updates over stale ids and over the SAME animator twice, palettes asked into a null pointer, frees of a rig / tracks
data / bone list that is in use, setters with out-of-range indices.  The contract: an error code; and the frame after it is the oracle's,
bit for bit (tests/test_anim_gpu.py::check_frame: poses, TRS, matrices, layer states, properties, root motion).

    python tools/fuzz_api_pose_gpu.py --first 0 --count 60 [--out gpurun_out/fuzz_api_pose_gpu.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fyrox_amd                      # noqa: E402
from fyrox_amd import _native         # noqa: E402
from fyrox_amd import anim as A       # noqa: E402
import anim_cases as cases            # noqa: E402
import oracle                         # noqa: E402  (the checker)
import test_anim_gpu as T             # noqa: E402

EARNED = {"FYX_ERR_INVALID_ARG", "FYX_ERR_UNKNOWN_ID", "FYX_ERR_UNSUPPORTED", "FYX_ERR_BONE_INDEX", "FYX_ERR_MISSING_ATTRIBUTE"}


def one_sequence(ctx, seed: int, stats: dict) -> None:
    rng = np.random.default_rng(seed + 99 * 10 ** 6)
    lib, h = _native.lib(), ctx._h
    members = []
    for k in range(int(rng.integers(1, 4))):
        sc = cases.random_machine(int(rng.integers(0, 10 ** 6)), n_bones=int(rng.integers(4, 12)), listy=bool(rng.integers(2)), lattice=bool(rng.integers(2)))
        n_inst = int(rng.integers(1, 4))
        o, p = cases.build_oracle(oracle, sc), cases.build_product(ctx, sc, n_inst)
        bones = list(range(sc.rig.n_nodes))
        A.create_bone_list(ctx, p.base_id + 50, p.base_id, bones)
        d_pal = ctx.malloc(n_inst * len(bones) * 64)
        p.set_palette_output(p.base_id + 50, d_pal.ptr)
        members.append((sc, o, p, n_inst, d_pal, bones))
    dt = members[0][0].dt
    stale = 987654321
    d_scratch = ctx.malloc(4096)

    def attempt(fn, *a, **kw):
        stats["calls"] += 1
        try:
            fn(*a, **kw)
            return True
        except fyrox_amd.FyxError as e:
            stats["errors"][e.status] = stats["errors"].get(e.status, 0) + 1
            if e.status not in EARNED:
                raise
            stats["refused"] += 1
            return False

    def raw(rc):
        stats["calls"] += 1
        if rc != 0:
            stats["refused"] += 1
            name = _native._STATUS_NAMES.get(rc, str(rc))
            stats["errors"][name] = stats["errors"].get(name, 0) + 1

    try:
        for f in range(int(rng.integers(6, 14))):
            # ---- abuse between the frames
            for _ in range(int(rng.integers(1, 6))):
                sc, o, p, n_inst, d_pal, bones = members[int(rng.integers(0, len(members)))]
                nn, na = sc.rig.n_nodes, len(sc.animations)
                which = int(rng.integers(0, 14))
                if which == 0: attempt(p.read, int(rng.choice([3, 15, 77, A.READ_ANIMATION_POSE + na + 5, A.READ_ANIMATION_BLEND_VIEW + na + 9, -1 & 0x7fffffff])))
                elif which == 1: attempt(p.set_palette_output, stale, d_pal.ptr)
                elif which == 2: attempt(p.palette, p.base_id + 50, 0)
                elif which == 3: attempt(p.palette, stale, d_pal.ptr)
                elif which == 4:
                    if attempt(p.set_skin_output, p.base_id + 50, stale, d_scratch.ptr, 0, 0):
                        raise AssertionError(f"seed {seed}: a skin output of a mesh that does not exist was accepted")
                elif which == 5: attempt(p.set_local_trs, int(rng.choice([nn, nn + 3, 2 ** 31 - 1])), np.zeros(10, np.float32))
                elif which == 6: attempt(p.set_local_trs, 0, np.zeros((n_inst + 2, 10), np.float32), int(rng.choice([0, n_inst])))
                elif which == 7: attempt(A.scene_update, ctx, [type("X", (), {"id": stale})()], dt)
                elif which == 8: raw(lib.fyx_rig_free(h, ctypes.c_uint64(p.base_id)))
                elif which == 9: raw(lib.fyx_tracks_data_free(h, ctypes.c_uint64(p.base_id + 1)))
                elif which == 10: raw(lib.fyx_bone_list_free(h, ctypes.c_uint64(p.base_id + 50)))
                elif which == 11: attempt(p.set_speed, int(rng.choice([na, na + 4, 2 ** 31 - 1])), 1.0)
                elif which == 12: attempt(p.set_layer_state, int(rng.choice([len(sc.machine.layers), 99])), 0, -1)
                else: attempt(p.blend_shape_weights, [int(rng.choice([999, 2 ** 31 - 1]))], [0.0], d_scratch.ptr)
            # ---- the frame: one by one, or as a scene (sometimes with a member listed twice: refused, then the scene as it should be)
            for sc, o, p, n_inst, d_pal, bones in members:
                for idx, par in sc.script.get(f, []):
                    o.set_parameter(idx, par)
                    p.set_parameter(idx, par)
                o.update_machine(dt) if sc.machine is not None else o.update_animations(dt)
            if rng.random() < 0.5:
                ps = [m[2] for m in members]
                if rng.random() < 0.4:
                    dup = ps + [ps[int(rng.integers(0, len(ps)))]]
                    if attempt(A.scene_update, ctx, dup, dt):
                        raise AssertionError(f"seed {seed}: a scene that lists an animator twice was accepted")
                A.scene_update(ctx, ps, dt)
                stats["calls"] += 1
            else:
                for sc, o, p, n_inst, d_pal, bones in members:
                    (p.update_machine if sc.machine is not None else p.update_animations)(dt)
                    stats["calls"] += 1
            for sc, o, p, n_inst, d_pal, bones in members:
                T.check_frame(p, o, sc, n_inst, f)
                pal = d_pal.download(np.float32, n_inst * len(bones) * 16).reshape(n_inst, len(bones), 16)
                assert T.same_bits(pal[0], o.palette(bones)).all(), f"seed {seed} frame {f}: palette"
                stats["frames_checked"] += 1
    finally:
        for sc, o, p, n_inst, d_pal, bones in members:
            o.close()
            p.free()
            d_pal.free()
        d_scratch.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=60)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    oracle.lib()
    stats = {"calls": 0, "errors": {}, "refused": 0, "frames_checked": 0}
    failed, t0 = [], time.time()
    with fyrox_amd.Context(0) as ctx:
        for seed in range(a.first, a.first + a.count):
            try:
                one_sequence(ctx, seed, stats)
            except Exception as e:     # noqa: BLE001
                import traceback
                failed.append({"seed": seed, "what": (str(e).strip().splitlines() or [repr(e)])[0][:300], "where": traceback.format_exc().strip().splitlines()[-3][:200]})
                print(json.dumps(failed[-1]), flush=True)
    rec = {"what": "random API sequences on the pose path (GPU): refused calls return codes, every frame behind them is the oracle's",
           "first_seed": a.first, "sequences": a.count, "calls": stats["calls"], "refused_calls": stats["refused"], "frames_checked_against_the_oracle": stats["frames_checked"],
           "error_codes_returned": stats["errors"], "failures": len(failed), "failed": failed, "seconds": round(time.time() - t0, 1)}
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
