import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, fyrox_amd
from fyrox_amd import anim as A, synth
ctx = fyrox_amd.Context(0)
seed = synth.SEED_BASE + 3
for depth in (8, 2, 32):
    rig = synth.make_rig(64, seed, chain_depth=depth)
    rid = 100 + depth
    A.create_rig(ctx, rid, rig)
    an = A.Animator(ctx, rid, rid, rig, 1000)
    for _ in range(20): an.update_transforms()
    ctx.sync(); ctx.timer_begin()
    for _ in range(200): an.update_transforms()
    print("chain_depth", depth, "update_transforms only: us", ctx.timer_end() * 1e3 / 200)
ctx.close()
