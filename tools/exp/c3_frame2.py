import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
mode = sys.argv[1]
if mode in ("torch", "torch_alloc"):
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(16, device="cuda")
    if mode == "torch_alloc":
        keep = [torch.empty(25_000_000, dtype=torch.float32, device="cuda") for _ in range(16)]
import fyrox_amd, bench
from fyrox_amd import synth
ctx = fyrox_amd.Context(0)
if mode == "meshes":
    mesh = synth.make_mesh(1_000_000, 256, synth.SEED_BASE + 4)
    for s in range(8):
        ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    d_pal = ctx.to_device(synth.make_palette(256, synth.SEED_BASE + 4))
    o = [ctx.malloc(12_000_064), ctx.malloc(12_000_064), ctx.malloc(16_000_064)]
    for i in range(3000):
        ctx.lbs_skin_device(i % 8, d_pal.ptr, 256, 1, o[0].ptr, o[1].ptr, o[2].ptr)
    ctx.sync()
sys.path.insert(0, os.path.join(ROOT, "tests"))
import anim_cases as cases
ctx.set_option("lbs.streams", 1)
r = bench._chain_record(ctx, "C3", cases.c5_blend_tree(n_bones=64, seed=synth.SEED_BASE + 3), synth.make_mesh(10_000, 64, synth.SEED_BASE + 3), 1000, 300, True, [0])
print(mode, {k: round(v, 4) for k, v in r.items() if k.startswith("frame_ms")}, flush=True)
