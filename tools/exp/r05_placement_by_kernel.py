#!/usr/bin/env python3
"""Round 5: choose the placement of a launch's outputs with the LAUNCH ITSELF as the probe (the fill-time probe does not predict:
profiles/r05_placement_pool/).  K candidate output triples (three separate allocations each) are allocated together, the real launch is
timed on each (per-dispatch events), the fastest are kept.  C4: 6 of 6 * K rotating sets; fused crowd: best of K.  Then the chosen sets
are measured again, against sets taken as they come.  One JSON line per leg."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def kernel_us(ctx, launch, n=200, warm=30):
    for _ in range(warm):
        launch()
    ctx.sync()
    ctx.set_option("lbs.timing", 1)
    ctx.kernel_time()
    for _ in range(n):
        launch()
    us, cnt = ctx.kernel_time()
    ctx.set_option("lbs.timing", 0)
    return us / max(cnt, 1)


with fyrox_amd.Context(0) as ctx:
    ctx.set_option("lbs.streams", 1)
    nv, nb, SETS = 1_000_000, 256, 6
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 4)
    pal = ctx.to_device(synth.make_palette(nb, synth.SEED_BASE + 4))
    for m in range(SETS):
        ctx.mesh_upload_soa(1 + m, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)

    def triple():
        return (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))

    def rotating(sets):
        st = {"k": 0}

        def rot():
            k = st["k"] % SETS
            st["k"] += 1
            o = sets[k]
            ctx.lbs_skin_device(1 + k, pal.ptr, nb, 1, o[0].ptr, o[1].ptr, o[2].ptr)
        return [round(kernel_us(ctx, rot, n=600), 2) for _ in range(3)]

    for rep in range(2):
        plain = [triple() for _ in range(SETS)]
        r_plain = rotating(plain)
        cands = [triple() for _ in range(SETS * K)]
        # probe: each candidate with the other streams of a CHOSEN-so-far rotation is not possible before choosing; probe alone, rotating inputs
        times = []
        for i, o in enumerate(cands):
            times.append(kernel_us(ctx, lambda o=o, i=i: ctx.lbs_skin_device(1 + i % SETS, pal.ptr, nb, 1, o[0].ptr, o[1].ptr, o[2].ptr), n=40, warm=6))
        order = np.argsort(times)
        chosen = [cands[i] for i in order[:SETS]]
        r_chosen = rotating(chosen)
        worst = [cands[i] for i in order[-SETS:]]
        r_worst = rotating(worst)
        print(json.dumps({"leg": "c4_lone_launch_rotating_6_sets", "candidates_per_set": K, "as_they_come_us": r_plain, "chosen_us": r_chosen, "rejected_slowest_us": r_worst,
                          "probe_us_sorted": [round(float(times[i]), 2) for i in order],
                          "frac_as_they_come": round(100e6 / (float(np.median(r_plain)) * 1e-6) / 8e12, 3), "frac_chosen": round(100e6 / (float(np.median(r_chosen)) * 1e-6) / 8e12, 3)}), flush=True)
        for o in plain + cands:
            for b in o:
                b.free()
    for m in range(SETS):
        ctx.mesh_free(1 + m)
    # fused crowd
    ni, nvc, nbc = 1000, 10_000, 64
    meshc = synth.make_mesh(nvc, nbc, synth.SEED_BASE + 3)
    pals = ctx.to_device(np.concatenate([synth.make_palette(nbc, synth.SEED_BASE + 3 + (i % 7)) for i in range(ni)]))
    ctx.mesh_upload_soa(20, meshc.pos, meshc.weights, meshc.indices, meshc.normal, meshc.tangent)
    unique = nvc * 60 + ni * nbc * 64 + ni * nvc * 40
    for exact in (0, 1):
        ctx.set_option("lbs.exact", exact)
        for rep in range(3):
            def ctriple():
                return (ctx.malloc(ni * nvc * 12 + 64), ctx.malloc(ni * nvc * 12 + 64), ctx.malloc(ni * nvc * 16 + 64))
            plain = ctriple()
            us_plain = kernel_us(ctx, lambda: ctx.lbs_skin_device(20, pals.ptr, nbc, ni, plain[0].ptr, plain[1].ptr, plain[2].ptr), n=100, warm=20)
            cands = [ctriple() for _ in range(K + 1)]
            probe = [kernel_us(ctx, lambda o=o: ctx.lbs_skin_device(20, pals.ptr, nbc, ni, o[0].ptr, o[1].ptr, o[2].ptr), n=12, warm=4) for o in cands]
            best = cands[int(np.argmin(probe))]
            us_best = kernel_us(ctx, lambda: ctx.lbs_skin_device(20, pals.ptr, nbc, ni, best[0].ptr, best[1].ptr, best[2].ptr), n=100, warm=20)
            print(json.dumps({"leg": "c3_crowd_exact" if exact else "c3_crowd_fused", "candidates": K + 1, "as_it_comes_us": round(us_plain, 2), "chosen_us": round(us_best, 2),
                              "probe_us": [round(x, 1) for x in probe], "frac_as_it_comes": round(unique / (us_plain * 1e-6) / 8e12, 3),
                              "frac_chosen": round(unique / (us_best * 1e-6) / 8e12, 3)}), flush=True)
            for o in [plain] + cands:
                for b in o:
                    b.free()
    ctx.set_option("lbs.exact", 1)
