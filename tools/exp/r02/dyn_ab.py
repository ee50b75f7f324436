#!/usr/bin/env python3
"""Experiment: two builds of the library in ONE process, measured alternately (A B A B ...) so that clock / thermal
drift between processes (about 1 us on a 19 us launch) does not decide the comparison.  C4 workload.
    python tools/exp/dyn_ab.py tools/exp/libs/libfyrox_hip_w5.so [rounds] >> gpurun_out/dyn_ab.jsonl"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def load(lib_path):
    for k in [k for k in sys.modules if k == "fyrox_amd" or k.startswith("fyrox_amd.")]:
        del sys.modules[k]
    if lib_path:
        os.environ["FYX_LIB_PATH"] = os.path.abspath(lib_path)
    else:
        os.environ.pop("FYX_LIB_PATH", None)
    return importlib.import_module("fyrox_amd")


variant = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mods = {"product": load(None), "variant": load(variant)}
from fyrox_amd import synth    # noqa: E402  (either copy: pure numpy)
NV, NB, SETS, N = 1_000_000, 256, 8, 400
m = synth.make_mesh(NV, NB, synth.SEED_BASE + 4)
pal_h = synth.make_palette(NB, synth.SEED_BASE + 4)
S = {}
for name, mod in mods.items():
    ctx = mod.Context(0)
    pal = ctx.to_device(pal_h)
    outs = []
    for s in range(SETS):
        ctx.mesh_upload_soa(s, m.pos, m.weights, m.indices, m.normal, m.tangent)
        outs.append((ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 12 + 64), ctx.malloc(NV * 16 + 64)))
    S[name] = (ctx, pal, outs)


def launch(name, i):
    ctx, pal, outs = S[name]
    s = i % SETS
    ctx.lbs_skin_device(s, pal.ptr, NB, 1, outs[s][0].ptr, outs[s][1].ptr, outs[s][2].ptr)


def snap(name):
    ctx, _, outs = S[name]
    ctx.sync()
    return [outs[0][k].download(np.uint32, NV * w) for k, w in ((0, 3), (1, 3), (2, 4))]


def kernel_us(name):
    ctx = S[name][0]
    ctx.set_option("lbs.streams", 1)
    for i in range(40):
        launch(name, i)
    ctx.set_option("lbs.timing", 1)
    ctx.kernel_time()
    for i in range(N):
        launch(name, i)
    us, n = ctx.kernel_time()
    ctx.set_option("lbs.timing", 0)
    return us / n


def overlapped_us(name):
    ctx = S[name][0]
    ctx.set_option("lbs.streams", 2)
    for i in range(40):
        launch(name, i)
    ctx.sync()
    ctx.timer_begin()
    for i in range(2 * N):
        launch(name, i)
    return ctx.timer_end() * 1e3 / (2 * N)


launch("product", 0); launch("variant", 0)
same = all(np.array_equal(a, b) for a, b in zip(snap("product"), snap("variant")))
res = {n: {"kernel_us": [], "overlapped_us": []} for n in S}
for r in range(rounds):
    for name in (("product", "variant") if r % 2 == 0 else ("variant", "product")):
        res[name]["kernel_us"].append(kernel_us(name))
    for name in (("product", "variant") if r % 2 == 0 else ("variant", "product")):
        res[name]["overlapped_us"].append(overlapped_us(name))
out = {"variant": os.path.basename(variant), "outputs_identical": same, "rounds": rounds}
for n in S:
    out[n] = {k: {"median": float(np.median(v)), "min": float(min(v)), "all": [round(x, 3) for x in v]} for k, v in res[n].items()}
print(json.dumps(out), flush=True)
