#!/usr/bin/env python3
"""Timeline of one default-variant skinning launch on the C4 workload (debug option lbs.probe): when the waves start,
when the palette is staged, when each wave issues / completes its last store.  GPU only; one JSON line."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fyrox_amd
from fyrox_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--launches", type=int, default=12)
args = ap.parse_args()
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
for kv in args.opt:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
NV, NB, SETS = 1_000_000, 256, 8
mesh = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
pal = ctx.to_device(synth.make_palette(NB, synth.SEED_BASE + 4))
outs = []
for s in range(SETS):
    ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))


def launch(i):
    s = i % SETS
    ctx.lbs_skin_device(s, pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)


for i in range(40):
    launch(i)
ctx.sync()
ctx.set_option("lbs.probe", 1)
bpc = ctx.get_option("lbs.blocks_per_cu")
n_waves = 256 * bpc * 8
rows = []
for i in range(args.launches):
    launch(i)
    launch(i + 1)          # the probed numbers are those of the LAST launch: it follows a launch, as in the bench
    ctx.sync()
    buf = np.zeros((n_waves, 4), np.uint64)
    ctx._check(ctx._l.fyx_debug_read_probe(ctx._h, buf.ctypes.data, n_waves))
    t = (buf & np.uint64((1 << 48) - 1)).astype(np.int64)    # the top bits carry the wave's hardware position (lbs_skin_dyn)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0           # 100 MHz ticks -> microseconds
    rows.append({"entry_last_us": float(us[:, 0].max()), "entry_p50_us": float(np.median(us[:, 0])),
                 "staged_p50_us": float(np.median(us[:, 1])), "staged_max_us": float(us[:, 1].max()),
                 "last_store_issued_p10_us": float(np.percentile(us[:, 2], 10)),
                 "last_store_issued_p50_us": float(np.median(us[:, 2])),
                 "last_store_issued_max_us": float(us[:, 2].max()),
                 "done_p50_us": float(np.median(us[:, 3])), "done_max_us": float(us[:, 3].max()),
                 "wave_busy_p50_us": float(np.median(us[:, 3] - us[:, 0]))})
# structure of the last launch's finish times: by XCD (block b runs on XCD b % 8), by address range (block index), by wave slot
blk = np.arange(n_waves) // 8
fin = us[:, 3]
by_xcd = [float(np.median(fin[(blk % 8) == x])) for x in range(8)]
by_range = [float(np.median(fin[(blk * 8 // (blk.max() + 1)) == k])) for k in range(8)]
by_wave = [float(np.median(fin[(np.arange(n_waves) % 8) == w])) for w in range(8)]
per_block_spread = float(np.median([fin[blk == b].max() - fin[blk == b].min() for b in range(0, blk.max() + 1, 7)]))
block_fin = np.array([fin[blk == b].max() for b in range(blk.max() + 1)])
structure = {"finish_median_by_xcd_us": by_xcd, "finish_median_by_address_eighth_us": by_range,
             "finish_median_by_wave_slot_us": by_wave, "median_spread_inside_a_block_us": per_block_spread,
             "block_finish_p10_p50_p90_max_us": [float(np.percentile(block_fin, q)) for q in (10, 50, 90, 100)]}
med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
print(json.dumps({"blocks_per_cu": bpc, "waves": n_waves, "median_over_launches": med, "structure_last_launch": structure}))
