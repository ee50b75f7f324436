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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
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
    for it in range(args.count):
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
