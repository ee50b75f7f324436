#!/usr/bin/env python3
"""Round 4 experiment: what a kernel on another stream costs the exact crowd skinning launch (C3: 1000 x 10 k vertices / 64 bones),
by what it does.  Per configuration: 40 skinning launches back to back on the context stream (lbs.streams = 1, lbs.timing = 1: each
launch's own duration), each followed at once by the co-runner launches on the co-runner's stream (tools/exp/r04_corunner.hip).
Prints the skinning kernel's mean duration, the period per iteration (events) and the co-runner's own duration."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fyrox_amd
from fyrox_amd import synth
co = ctypes.CDLL(os.path.join(ROOT, "tools/exp/libs/libcorunner.so"))
co.corun_last_ms.restype = ctypes.c_float
N = 1000
ctx = fyrox_amd.Context(0)
ctx.set_option("lbs.streams", 1)
ctx.set_option("lbs.exact", int(os.environ.get("EXACT", "1")))
assert co.corun_init() == 0
seed = synth.SEED_BASE + 3
mesh = synth.make_mesh(10_000, 64, seed)
ctx.mesh_upload_soa(3, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
pal = ctx.to_device(synth.make_palette(64, seed, n_instances=N))
nv = 10_000 * N
outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
skin = ctx._l.fyx_lbs_skin_device
SK = (ctx._h, ctypes.c_uint64(3), ctypes.c_void_p(pal.ptr), ctypes.c_uint32(64), ctypes.c_uint32(N), ctypes.c_void_p(outs[0].ptr), ctypes.c_void_p(outs[1].ptr), ctypes.c_void_p(outs[2].ptr))
# (name, [(kind, regs, grid, us, hops), ...])
CONFIGS = [
    ("none", []),
    ("scatter: 3072 x 4 waves, one scattered 16-byte store per lane (786 k partial lines, 12.6 MB)", [(2, 56, 3072, 0, 1)]),
    ("records: 3072 x 4 waves, three adjacent lanes write one scattered 48-byte record (262 k records, 12.6 MB)", [(4, 56, 3072, 0, 1)]),
    ("scatter: 1024 x 4 waves, one scattered store per lane (262 k partial lines, 4.2 MB)", [(2, 56, 1024, 0, 1)]),
    ("occupy: 3072 x 4 waves of 56 VGPRs sleeping 1 us", [(0, 56, 3072, 1, 0)]),
]
ctx.set_option("lbs.timing", 1)
def run(cos, iters):
    for _ in range(iters):
        ctx._check(skin(*SK))
        for (kind, regs, grid, us, hops) in cos:
            assert co.corun_launch(kind, regs, grid, us, hops) == 0


run([], 300)       # clocks up
res = {name: {"skin_kernel_us": [], "period_us": [], "co_runner_us": []} for name, _ in CONFIGS}
tot, n = ctypes.c_double(), ctypes.c_uint32()
for rnd in range(3):
    for name, cos in CONFIGS:
        run(cos, 20)
        ctx.sync(); co.corun_sync()
        ctx._check(ctx._l.fyx_debug_kernel_time(ctx._h, ctypes.byref(tot), ctypes.byref(n)))
        ctx.timer_begin()
        run(cos, 60)
        period = ctx.timer_end() / 60 * 1e3
        co.corun_sync()
        ctx._check(ctx._l.fyx_debug_kernel_time(ctx._h, ctypes.byref(tot), ctypes.byref(n)))
        res[name]["skin_kernel_us"].append(round(tot.value / max(n.value, 1), 1))
        res[name]["period_us"].append(round(period, 1))
        res[name]["co_runner_us"].append(round(co.corun_last_ms() * 1e3, 1) if cos else 0.0)
for name, _ in CONFIGS:
    print(json.dumps({"co_runner": name, **res[name]}), flush=True)
ctx.close()
