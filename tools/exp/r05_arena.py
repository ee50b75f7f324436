#!/usr/bin/env python3
"""Round 5: the lone C4 launch rotating over 12 sets (1.2 GB walked over), every stream of every set carved from ONE arena allocated at
process start (fyx_lbs_skin_streams on raw pointers) against the same streams in separate allocations: is the footprint effect
(tools/exp/r05_sets_sweep.py) a matter of how the memory was allocated?  HIP events around back-to-back launches on one stream."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fyrox_amd
from fyrox_amd import synth

NV, NB, MAXS = 1_000_000, 256, 12
SIZES = [NV * 12, NV * 12, NV * 16, NV * 16, NV * 4, NV * 12, NV * 12, NV * 16]      # pos nrm tan wgt idx | out pos nrm tan


def period_us(ctx, launch, n=600, warm=40):
    for i in range(warm):
        launch(i)
    ctx.sync()
    ctx.timer_begin()
    for i in range(n):
        launch(i)
    return ctx.timer_end() * 1e3 / n


with fyrox_amd.Context(0) as ctx:
    ctx.set_option("lbs.streams", 1)
    align = int(sys.argv[1]) if len(sys.argv) > 1 else (2 << 20)
    per_set = sum((s + align - 1) // align * align for s in SIZES)
    arena = ctx.malloc(per_set * MAXS + align)          # FIRST allocation of the process
    base = (arena.ptr + align - 1) // align * align
    mesh = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
    pal = ctx.to_device(synth.make_palette(NB, synth.SEED_BASE + 4))
    host = [np.ascontiguousarray(mesh.pos, np.float32), np.ascontiguousarray(mesh.normal, np.float32), np.ascontiguousarray(mesh.tangent, np.float32),
            np.ascontiguousarray(mesh.weights, np.float32), np.ascontiguousarray(mesh.indices).view(np.uint8)]
    import ctypes

    def put(dst, arr):
        ctx._check(ctx._l.fyx_memcpy_h2d(ctx._h, dst, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))

    arena_sets, sep_sets = [], []
    for s in range(MAXS):
        p, ptrs = base + s * per_set, []
        for sz in SIZES:
            ptrs.append(p)
            p += (sz + align - 1) // align * align
        for k in range(5):
            put(ptrs[k], host[k])
        arena_sets.append(ptrs)
    for s in range(MAXS):
        bufs = [ctx.malloc(sz + 64) for sz in SIZES]
        for k in range(5):
            put(bufs[k].ptr, host[k])
        sep_sets.append([b.ptr for b in bufs])
    ctx.sync()

    def make(sets, k):
        def launch(i):
            q = sets[i % k]
            ctx.lbs_skin_streams(NV, q[0], q[1], q[2], q[3], q[4], pal.ptr, NB, 1, q[5], q[6], q[7])
        return launch
    out = {"align": align, "arena": {}, "separate": {}}
    for k in (1, 4, 6, 8, 12, 8, 12):
        out["arena"].setdefault(str(k), []).append(round(period_us(ctx, make(arena_sets, k)), 2))
        out["separate"].setdefault(str(k), []).append(round(period_us(ctx, make(sep_sets, k)), 2))
    print(json.dumps(out), flush=True)
