#!/usr/bin/env python3
"""Bug hunt on the GPU box, skinning side: random mesh sizes (ragged, tiny, one vertex, a few hundred thousand), bone counts 1 .. 256,
instance counts, SoA / AnimatedVertex uploads and every launch option of the LBS kernels -- each output compared bit for bit with the
oracle's serial loop (lbs.exact = 1) or within 1e-5 (lbs.exact = 0).

    python tools/fuzz_lbs_gpu.py --count 300 [--seed 1] [--out gpurun_out/fuzz_lbs.json]

Test infrastructure: the oracle is the checker here exactly as in tests/."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def fuzz_ex(T, synth, orc, ctx, rng, it):
    """One fyx_lbs_skin_ex launch over a random vertex layout; returns its configuration (raises with .cfg on a mismatch)."""
    nv = int(rng.choice([1, 63, 64, 65, 129, int(rng.integers(1, 3000)), int(rng.integers(3000, 60_000))]))
    nb = int(rng.choice([1, 4, 24, 64, 200, 256]))
    n_inst = int(rng.choice([1, 1, 2, 3]))
    shapes = int(rng.choice([0, 0, 1, 3, 8]))
    exact = int(rng.random() < 0.8)
    # input layout: attributes placed one after another in random order with random 4-byte gaps
    have_n, have_t = bool(rng.random() < 0.8), bool(rng.random() < 0.7)
    parts = [("pos", 12), ("weights", 16), ("indices", 4)] + ([("normal", 12)] if have_n else []) + ([("tangent", 16)] if have_t else [])
    order = [parts[k] for k in rng.permutation(len(parts))]
    offs, at = {}, 0
    for key, size in order:
        at += 4 * int(rng.integers(0, 3))
        offs[key] = at
        at += size
    stride = at + 4 * int(rng.integers(0, 4))
    if stride > 160:
        stride = at
    out_mode = int(rng.integers(0, 3))     # 0: the mesh's own layout out (out_stride = 0), 1: another interleaved layout, 2: three streams
    cfg = {"verts": nv, "bones": nb, "instances": n_inst, "shapes": shapes, "lbs.exact": exact, "stride": stride, "offsets": offs, "out_mode": out_mode}
    try:
        import numpy as np
        seed = synth.SEED_BASE + 5000 + it
        m = synth.make_mesh(nv, nb, seed, coherent=bool(rng.integers(2)))
        pal = synth.make_palette(nb, seed, n_instances=n_inst)
        src = T._custom_aos(m, stride, offs)
        ctx.mesh_upload(901, src.reshape(-1), nv, stride, off_pos=offs["pos"], off_normal=offs.get("normal", -1), off_tangent=offs.get("tangent", -1),
                        off_weights=offs["weights"], off_indices=offs["indices"])
        mn, mt = (m.normal if have_n else None), (m.tangent if have_t else None)
        weights = None
        if shapes:
            storage, plane, w = synth.make_blend_shapes(nv, shapes, seed)
            ctx.mesh_set_blend_shapes(901, storage, shapes, plane)
            weights = np.stack([w * np.float32(1.0 - 0.3 * i) for i in range(n_inst)])
        ref = {"pos": [], "normal": [], "tangent": []}
        for i in range(n_inst):
            p_, n_, t_ = m.pos, mn, mt
            if shapes:
                p_, n_, t_ = orc.apply_blend_shapes(m.pos, m.normal if have_n else np.zeros_like(m.pos), m.tangent if have_t else np.zeros((nv, 4), np.float32),
                                                    storage, plane, weights[i])
                n_, t_ = (n_ if have_n else None), (t_ if have_t else None)
            r = orc.lbs_skin(p_, m.weights, m.indices, pal[i * nb:(i + 1) * nb], n_, t_, threads=0)
            for k in ref:
                if k in r and r[k] is not None:
                    ref[k].append(r[k])
        ref = {k: np.concatenate(v) for k, v in ref.items() if v}
        ctx.set_option("lbs.exact", exact)

        def compare(got, want, key):
            if exact:
                assert np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.ascontiguousarray(want).view(np.uint32)), f"{key}: max rel err {T.rel_err(got, want):.3e}"
            else:
                assert T.rel_err(got, want) <= T.REL_TOL, f"{key}: {T.rel_err(got, want):.3e}"

        d_pal = ctx.to_device(pal)
        d_w = ctx.to_device(weights) if shapes else None
        guard = 256
        try:
            if out_mode == 2:
                bufs = [ctx.to_device(np.full(nv * n_inst * w_ * 4 + guard, 0xEE, np.uint8)) for w_ in (3, 3, 4)]
                ctx.lbs_skin_ex(901, d_pal.ptr, nb, n_inst, d_blend_shape_weights=d_w.ptr if d_w else 0, n_blend_shapes=shapes,
                                d_out_pos=bufs[0].ptr, d_out_normal=bufs[1].ptr if have_n else 0, d_out_tangent=bufs[2].ptr if have_t else 0)
                ctx.sync()
                for b, key, w_ in zip(bufs, ("pos", "normal", "tangent"), (3, 3, 4)):
                    raw = b.download(np.uint8, nv * n_inst * w_ * 4 + guard)
                    assert np.all(raw[-guard:] == 0xEE), f"{key}: wrote past the end"
                    if key in ref:
                        compare(raw[:-guard].view(np.float32).reshape(-1, w_), ref[key], key)
                    else:
                        assert np.all(raw == 0xEE), f"{key}: an output that does not exist was written"
                    b.free()
            else:
                if out_mode == 0:
                    o_stride, o_offs, sizes = stride, offs, {"pos": 12, "normal": 12, "tangent": 12}
                    init = np.full((nv * n_inst, stride), 0xEE, np.uint8)
                    expect_rest = np.tile(src, (n_inst, 1))
                else:
                    o_parts = [("pos", 12)] + ([("normal", 12)] if have_n and rng.random() < 0.8 else []) + ([("tangent", 16)] if have_t and rng.random() < 0.8 else [])
                    o_offs, at = {}, 0
                    for k in rng.permutation(len(o_parts)):
                        at += 4 * int(rng.integers(0, 3))
                        o_offs[o_parts[k][0]] = at
                        at += o_parts[k][1]
                    o_stride = at + 4 * int(rng.integers(0, 5))
                    sizes = {"pos": 12, "normal": 12, "tangent": 16}
                    init = np.full((nv * n_inst, o_stride), 0xA5, np.uint8)
                    expect_rest = init
                    cfg["out_stride"], cfg["out_offsets"] = o_stride, o_offs
                buf = ctx.to_device(np.concatenate([init.reshape(-1), np.full(guard, 0xEE, np.uint8)]))
                ctx.lbs_skin_ex(901, d_pal.ptr, nb, n_inst, d_blend_shape_weights=d_w.ptr if d_w else 0, n_blend_shapes=shapes, d_out_vertices=buf.ptr,
                                out_stride=0 if out_mode == 0 else o_stride, out_off_pos=o_offs.get("pos", -1) if out_mode else -1,
                                out_off_normal=o_offs.get("normal", -1) if out_mode else -1, out_off_tangent=o_offs.get("tangent", -1) if out_mode else -1)
                ctx.sync()
                raw = buf.download(np.uint8, init.size + guard)
                buf.free()
                assert np.all(raw[-guard:] == 0xEE), "wrote past the last vertex"
                raw = raw[:-guard].reshape(nv * n_inst, o_stride)
                touched = np.zeros(o_stride, bool)
                for key, size in sizes.items():
                    if key in o_offs and key in ref:
                        got = np.ascontiguousarray(raw[:, o_offs[key]:o_offs[key] + size]).view(np.float32)
                        compare(got, ref[key][:, :size // 4], key)
                        touched[o_offs[key]:o_offs[key] + size] = True
                assert np.array_equal(raw[:, ~touched], expect_rest[:, ~touched]), "bytes outside the written attributes changed"
        finally:
            d_pal.free()
            if d_w:
                d_w.free()
    except Exception as e:   # noqa: BLE001
        e.cfg = cfg
        raise
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ex", action="store_true", help="fyx_lbs_skin_ex: random vertex layouts in -> out, blend shapes, interleaved outputs")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import fyrox_amd
    import oracle
    from fyrox_amd import synth
    import test_lbs_gpu as T

    oracle.lib()
    ctx = fyrox_amd.Context(0)
    rng = np.random.default_rng(args.seed)
    fails, t0 = [], time.time()
    for it in range(args.count if args.ex else 0):
        cfg = {}
        try:
            T._set_defaults(ctx)
            cfg = fuzz_ex(T, synth, oracle, ctx, rng, it)
        except Exception as e:   # noqa: BLE001
            fails.append({"iteration": it, "config": getattr(e, "cfg", cfg), "error": (str(e).strip().splitlines() or [repr(e)])[0][:300]})
        finally:
            try:
                ctx.mesh_free(901)
            except Exception:   # noqa: BLE001
                pass
    for it in range(0 if args.ex else args.count):
        r = rng.random()
        nv = int(rng.integers(1, 70) if r < 0.15 else rng.integers(1, 5000) if r < 0.6 else rng.integers(5000, 300_000))
        nb = int(rng.choice([1, 2, 3, 4, 7, 16, 33, 64, 100, 128, 200, 255, 256]))
        n_inst = int(rng.choice([1, 1, 1, 2, 3, 5, 17, 40]))
        if nv * n_inst > 1_500_000:
            n_inst = 1
        cfg = {"verts": nv, "bones": nb, "instances": n_inst, "coherent": bool(rng.integers(2)), "aos": bool(rng.integers(2)),
               "lbs.exact": int(rng.random() < 0.8), "lbs.dyn": int(rng.integers(2)), "lbs.blocks_per_cu": int(rng.choice([1, 2, 4, 8, 16])),
               "lbs.crowd": int(rng.choice([-1, 0, 1])), "lbs.crowd_ipb": int(rng.choice([0, 0, 1, 2, 3, 8, 16])),
               "lbs.crowd_lean": int(rng.integers(2)), "lbs.streams": int(rng.choice([1, 2])), "aabb": bool(rng.integers(2)) and n_inst == 1}
        try:
            T._set_defaults(ctx)
            for k, v in cfg.items():
                if k.startswith("lbs."):
                    ctx.set_option(k, v)
            seed = synth.SEED_BASE + 100 + it
            m = synth.make_mesh(nv, nb, seed, cfg["coherent"])
            pal = synth.make_palette(nb, seed, n_instances=n_inst)
            T.upload(ctx, 900, m, cfg["aos"])
            got = ctx.lbs_skin(900, pal, n_instances=n_inst, aabb=True) if cfg["aabb"] else ctx.lbs_skin(900, pal, n_instances=n_inst)
            ref = T.oracle_skin(oracle, m, pal, n_inst)
            if cfg["lbs.exact"]:
                T.assert_bit_exact(got, ref)
            else:
                for k in ("pos", "normal", "tangent"):
                    assert T.rel_err(got[k], ref[k]) <= T.REL_TOL, f"{k}: {T.rel_err(got[k], ref[k]):.3e}"
            if cfg["aabb"] and cfg["lbs.exact"]:
                assert np.array_equal(got["aabb"][:3], ref["pos"].min(axis=0)) and np.array_equal(got["aabb"][3:], ref["pos"].max(axis=0)), "aabb"
        except Exception as e:   # noqa: BLE001 -- every failure is a finding
            fails.append({"iteration": it, "config": cfg, "error": (str(e).strip().splitlines() or [repr(e)])[0][:300]})
        finally:
            try:
                ctx.mesh_free(900)
            except Exception:   # noqa: BLE001
                pass
    T._set_defaults(ctx)
    rec = {"what": "random LBS launches on the GPU against the oracle", "seed": args.seed, "launches": args.count, "failures": len(fails),
           "failed": fails[:40], "seconds": round(time.time() - t0, 1)}
    line = json.dumps(rec)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
