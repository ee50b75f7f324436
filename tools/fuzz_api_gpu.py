#!/usr/bin/env python3
"""API-sequence fuzz of the DATA path on the GPU box: random sequences of mesh uploads, frees, skinning calls of every entry point, option
changes and joins, in which every argument the library can check is sometimes wrong -- unknown and freed mesh ids, palettes shorter than
the mesh's largest bone index, null palettes and outputs, zero instances, normals asked of a mesh without them, interleaved outputs that do
not fit their stride, blend-shape weights without shapes, batches holding one bad job.  (Buffer SIZES are the caller's word in a C ABI: every
buffer here is as large as the call says.)  The contract: an error code and nothing written; and the library's state survives -- after every
refused call a good call on the same context must give the oracle's bits.  Small meshes, a sync after every sequence.

    python tools/fuzz_api_gpu.py --first 0 --count 150 [--out gpurun_out/fuzz_api_gpu.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import fyrox_amd                      # noqa: E402
from fyrox_amd import synth           # noqa: E402
import oracle                         # noqa: E402  (the checker)

EARNED = {"FYX_ERR_INVALID_ARG", "FYX_ERR_UNKNOWN_ID", "FYX_ERR_UNSUPPORTED", "FYX_ERR_BONE_INDEX", "FYX_ERR_MISSING_ATTRIBUTE"}
L = synth.ANIMATED_VERTEX


def one_sequence(ctx, seed: int, stats: dict) -> None:
    rng = np.random.default_rng(seed + 88 * 10 ** 6)
    meshes = {}                        # id -> (mesh, has_normal, has_tangent, aos)
    stale = [4242424242]
    GUARD = 0x7FC0DEAD

    def attempt(fn, *a, **kw):
        stats["calls"] += 1
        try:
            fn(*a, **kw)
            return True
        except fyrox_amd.FyxError as e:
            stats["errors"][e.status] = stats["errors"].get(e.status, 0) + 1
            if e.status not in EARNED:
                raise
            return False

    def new_mesh():
        mid = int(rng.integers(1, 40))
        n, nb = int(rng.choice([1, 63, 64, 65, 777, 4097, 20_000])), int(rng.integers(1, 65))
        m = synth.make_mesh(n, nb, int(rng.integers(0, 10 ** 6)), coherent=bool(rng.integers(2)))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            ctx.mesh_upload_soa(mid, m.pos, m.weights, m.indices, m.normal, m.tangent)
            meshes[mid] = (m, True, True, False)
        elif kind == 1:
            ctx.mesh_upload_soa(mid, m.pos, m.weights, m.indices, None, None)
            meshes[mid] = (m, False, False, False)
        else:
            ctx.mesh_upload(mid, m.to_animated_vertex_aos(), n, L["stride"], off_pos=L["off_pos"], off_normal=L["off_normal"], off_tangent=L["off_tangent"],
                            off_weights=L["off_weights"], off_indices=L["off_indices"])
            meshes[mid] = (m, True, True, True)
        stats["calls"] += 1
        return mid

    def good_call(mid):
        """a call that must work, checked against the oracle: the state survived whatever was refused before"""
        m, hn, ht, _ = meshes[mid]
        nb = int(m.indices.max()) + 1
        pal = synth.make_palette(nb, int(rng.integers(0, 10 ** 6)))
        ref = oracle.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal if hn else None, m.tangent if ht else None, threads=0)
        got = ctx.lbs_skin(mid, pal, want=tuple(k for k, on in (("pos", True), ("normal", hn), ("tangent", ht)) if on))
        stats["calls"] += 1
        for k in got:
            assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), f"seed {seed}: good call on mesh {mid}: {k} differs from the oracle"
        stats["good_calls_checked"] += 1

    for step in range(int(rng.integers(15, 40))):
        r = rng.random()
        if r < 0.2 or not meshes:
            new_mesh()
            continue
        mid = int(rng.choice(list(meshes)))
        m, hn, ht, aos = meshes[mid]
        n, nb = m.n_verts, int(m.indices.max()) + 1
        ni = int(rng.choice([1, 1, 2, 5]))
        d_pal = ctx.to_device(synth.make_palette(nb, 5, n_instances=ni))
        outs = [ctx.malloc((n * ni * w + 16) * 4) for w in (3, 3, 4)]
        for b, w in zip(outs, (3, 3, 4)):
            b.upload(np.full(n * ni * w + 16, GUARD, np.uint32))
        refused = False
        try:
            which = int(rng.integers(0, 12))
            if which == 0:                                               # unknown / freed id
                refused = not attempt(ctx.lbs_skin_device, int(rng.choice(stale)), d_pal.ptr, nb, ni, outs[0].ptr, 0, 0)
            elif which == 1:                                             # palette too short (one short, or much)
                short = int(rng.choice([nb - 1, 0, max(nb // 2, 0)]))
                refused = not attempt(ctx.lbs_skin_device, mid, d_pal.ptr, short, ni, outs[0].ptr, 0, 0)
            elif which == 2:                                             # null palette
                refused = not attempt(ctx.lbs_skin_device, mid, 0, nb, ni, outs[0].ptr, 0, 0)
            elif which == 3:                                             # zero instances / nothing asked for: allowed no-ops or refusals, nothing written
                refused = not attempt(ctx.lbs_skin_device, mid, d_pal.ptr, nb, int(rng.choice([0, ni])), 0, 0, 0) or True
            elif which == 4:                                             # an attribute the mesh does not have
                refused = not attempt(ctx.lbs_skin_device, mid, d_pal.ptr, nb, ni, outs[0].ptr, outs[1].ptr, outs[2].ptr) if not hn else False
            elif which == 5:                                             # more bones than a palette may hold
                refused = not attempt(ctx.lbs_skin_device, mid, d_pal.ptr, int(rng.choice([257, 1000, 2 ** 31 - 1])), ni, outs[0].ptr, 0, 0)
            elif which == 6:                                             # interleaved output that does not fit its stride / misaligned offsets
                refused = not attempt(ctx.lbs_skin_ex, mid, d_pal.ptr, nb, ni, d_out_vertices=outs[2].ptr, out_stride=int(rng.choice([8, 12])),
                                      out_off_pos=int(rng.choice([2, 4, 999])), out_off_normal=-1, out_off_tangent=-1)      # (every combination is invalid: nothing may be written)
            elif which == 7:                                             # blend-shape weights for a mesh without shapes
                refused = not attempt(ctx.lbs_skin_ex, mid, d_pal.ptr, nb, ni, d_blend_shape_weights=outs[1].ptr, n_blend_shapes=int(rng.choice([1, 4, 64])), d_out_pos=outs[0].ptr)
            elif which == 8:                                             # vertex-buffer out of a mesh uploaded as streams
                refused = not attempt(ctx.lbs_skin_ex, mid, d_pal.ptr, nb, ni, d_out_vertices=outs[2].ptr, out_stride=0) if not aos else False
            elif which == 9:                                             # a batch with one bad job: all or nothing
                other = int(rng.choice(list(meshes)))
                bad = (int(rng.choice(stale)), d_pal.ptr, nb, 1, outs[0].ptr, 0, 0) if rng.random() < 0.5 else (other, d_pal.ptr, 0, 1, outs[0].ptr, 0, 0)
                refused = not attempt(ctx.lbs_skin_batch, [(mid, d_pal.ptr, nb, ni, outs[0].ptr, 0, 0), bad])
            elif which == 10:                                            # AABB calls with bad arguments
                refused = not attempt(ctx.skinned_aabb_device, int(rng.choice(stale + [mid])), d_pal.ptr, int(rng.choice([nb, nb - 1, 0])), ni, int(rng.choice([0, outs[0].ptr])))
            else:                                                        # options out of range
                key, v = [("lbs.blocks_per_cu", 0), ("lbs.blocks_per_cu", 65), ("lbs.streams", 0), ("lbs.streams", 99), ("lbs.crowd", 7), ("no.such.option", 1)][int(rng.integers(0, 6))]
                refused = not attempt(ctx.set_option, key, v)
            ctx.sync()
            if refused:
                stats["refused"] += 1
                for b, w in zip(outs, (3, 3, 4)):                        # a refused call wrote nothing
                    raw = b.download(np.uint32, n * ni * w + 16)
                    assert (raw == GUARD).all(), f"seed {seed} step {step}: a refused call (kind {which}) wrote to an output"
                good_call(mid)
        finally:
            d_pal.free()
            for b in outs:
                b.free()
        if rng.random() < 0.15:
            victim = int(rng.choice(list(meshes)))
            ctx.mesh_free(victim)
            del meshes[victim]
            stale.append(victim)
            stats["calls"] += 1
            attempt(ctx.mesh_free, victim)                               # twice
    for mid in list(meshes):
        ctx.mesh_free(mid)
    ctx.sync()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    oracle.lib()
    stats = {"calls": 0, "errors": {}, "refused": 0, "good_calls_checked": 0}
    failed, t0 = [], time.time()
    with fyrox_amd.Context(0) as ctx:
        for seed in range(a.first, a.first + a.count):
            try:
                one_sequence(ctx, seed, stats)
            except Exception as e:     # noqa: BLE001
                failed.append({"seed": seed, "what": (str(e).strip().splitlines() or [repr(e)])[0][:300]})
                print(json.dumps(failed[-1]), flush=True)
    rec = {"what": "random API sequences on the data path (GPU): refused calls return codes and write nothing, the next good call gives the oracle's bits",
           "first_seed": a.first, "sequences": a.count, "calls": stats["calls"], "refused_calls": stats["refused"], "good_calls_checked_against_the_oracle": stats["good_calls_checked"],
           "error_codes_returned": stats["errors"], "failures": len(failed), "failed": failed, "seconds": round(time.time() - t0, 1)}
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
