#!/usr/bin/env python3
"""A short serialized run for two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; bench.py runs them itself for roofline.traffic):
N launches of the calibration stream copy (known bytes: 60 MB read + 40 MB written per launch), then N launches of the headline
(C4: 1 M vertices / 256 bones, lbs_skin_dyn), over rotating buffer sets.  python tools/pmc_probe_headline.py [N] [verts] [bones]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fyrox_amd
from fyrox_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 256
SETS, UNITS = 4, 1_250_000
ctx = fyrox_amd.Context(int(os.environ.get("FYX_BENCH_DEVICE", "0")))
ctx.set_option("lbs.streams", 1)
d_pal = ctx.to_device(synth.make_palette(NB, synth.SEED_BASE + 4))
m = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
outs, srcs, dsts = [], [], []
for s in range(SETS):
    ctx.mesh_upload_soa(10 + s, m.pos, m.weights, m.indices, m.normal, m.tangent)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))
    srcs.append(ctx.to_device(np.full(UNITS * 12, np.float32(s + 1))))
    dsts.append(ctx.malloc(UNITS * 32))
for i in range(N):
    ctx.calib_stream_copy(srcs[i % SETS].ptr, dsts[i % SETS].ptr, UNITS)
ctx.sync()
for i in range(N):
    s = i % SETS
    ctx.lbs_skin_device(10 + s, d_pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)
ctx.sync()
ctx.close()
