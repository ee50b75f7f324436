#!/usr/bin/env python3
"""Experiment: lbs_skin_dyn at another register budget (FYX_EXP_DYN_WAVES builds, tools/exp/build_variants.sh), C4 workload.
The kernel alone (per-dispatch events, one stream), launch to launch on one stream, overlapped on two streams, and the
output compared bit for bit with lbs_skin's.
    FYX_LIB_PATH=tools/exp/libs/libfyrox_hip_w5.so python tools/exp/dyn_waves.py w5 >> gpurun_out/dyn_waves.jsonl"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fyrox_amd
from fyrox_amd import synth

tag = sys.argv[1] if len(sys.argv) > 1 else "product"
NV, NB, SETS, N = 1_000_000, 256, 8, 600
cache = "/tmp/fyx_c4_mesh.npz"
if os.path.exists(cache):
    z = np.load(cache)
    pos, wgt, idx, nrm, tan = (z[k] for k in ("pos", "wgt", "idx", "nrm", "tan"))
else:
    m = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
    pos, wgt, idx, nrm, tan = m.pos, m.weights, m.indices, m.normal, m.tangent
    np.savez(cache, pos=pos, wgt=wgt, idx=idx, nrm=nrm, tan=tan)
ctx = fyrox_amd.Context(0)
extra = [kv.split("=") for kv in sys.argv[2:]]      # options applied before the measurement (lbs.dyn_bpc=4 ...)
streams = 2
for k, v in extra:
    if k == "lbs.streams":
        streams = int(v)
    else:
        ctx.set_option(k, int(v))
pal = ctx.to_device(synth.make_palette(NB, synth.SEED_BASE + 4))
outs = []
for s in range(SETS):
    ctx.mesh_upload_soa(s, pos, wgt, idx, nrm, tan)
    outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))


def launch(i):
    s = i % SETS
    ctx.lbs_skin_device(s, pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)


def snap():
    ctx.sync()
    return [outs[0][k].download(np.uint32, NV * w) for k, w in ((0, 3), (1, 3), (2, 4))]


def period(n):
    ctx.sync()
    ctx.timer_begin()
    for i in range(n):
        launch(i)
    return ctx.timer_end() * 1e3 / n


ctx.set_option("lbs.dyn", 0)
launch(0)
ref = snap()
ctx.set_option("lbs.dyn", 1)
for b in outs[0]:
    b.upload(np.zeros(b.nbytes // 4, np.uint32))
launch(0)
same = all(np.array_equal(a, b) for a, b in zip(ref, snap()))
ctx.set_option("lbs.streams", 1)
for i in range(60):
    launch(i)
ser = [period(N) for _ in range(3)]
ctx.set_option("lbs.timing", 1)
ctx.kernel_time()
ker = []
for _ in range(3):
    for i in range(N):
        launch(i)
    us, n = ctx.kernel_time()
    ker.append(us / n)
ctx.set_option("lbs.timing", 0)
ctx.set_option("lbs.streams", streams)
for i in range(60):
    launch(i)
ovl = [period(2 * N) for _ in range(3)]
b = 100.0 * NV
print(json.dumps({"tag": tag, "options": sys.argv[2:], "bit_identical_to_lbs_skin": same, "kernel_us": float(np.median(ker)), "kernel_us_all": ker,
                  "frac": b / (float(np.median(ker)) * 1e-6) / 8e12, "serialized_period_us": float(np.median(ser)),
                  "overlapped_us": float(np.median(ovl)), "overlapped_frac": b / (float(np.median(ovl)) * 1e-6) / 8e12}), flush=True)
ctx.close()
