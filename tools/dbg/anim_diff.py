import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import fyrox_amd, oracle
from fyrox_amd import anim as A
import anim_cases as cases
ctx = fyrox_amd.Context(0)
for kind in (0, 1, 2):
    sc = cases.player_only(euler_every=10**9, key_kind=kind)
    o = cases.build_oracle(oracle, sc); p = cases.build_product(ctx, sc, 1)
    for f in range(3):
        o.update_animations(sc.dt); p.update_animations(sc.dt)
        got = p.read(A.READ_ANIMATION_POSE)[0]; ref = o.animation_pose(0)
        d = got.view(np.uint32) != ref.view(np.uint32)
        print("kind", kind, "frame", f, "mismatch per column", d.sum(axis=0))
        if d.any():
            n, c = np.argwhere(d)[0]
            print("  node", n, "col", c, got[n, c], ref[n, c], got[n].tolist(), ref[n].tolist())
