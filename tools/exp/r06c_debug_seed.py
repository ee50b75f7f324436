#!/usr/bin/env python3
"""Re-run one fuzz seed of tools/fuzz_gpu.py with the values printed where a bit-exact comparison fails.
    python tools/exp/r06c_debug_seed.py 51202 --subnormal --listy --lattice --edits --bones 8"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import test_anim_gpu as T
orig = T.check
def check(got, ref, exact, what):
    try:
        orig(got, ref, exact, what)
    except AssertionError:
        g, r = np.ascontiguousarray(got), np.ascontiguousarray(ref)
        bad = g.view(np.uint32) != r.view(np.uint32)
        print("MISMATCH", what, "at", np.argwhere(bad)[:8].tolist())
        print(" got", g[bad][:8].tolist(), [hex(x) for x in g.view(np.uint32)[bad][:8].tolist()])
        print(" ref", r[bad][:8].tolist(), [hex(x) for x in r.view(np.uint32)[bad][:8].tolist()])
        print(" got all", g.ravel()[:16].tolist()); print(" ref all", r.ravel()[:16].tolist())
        raise
T.check = check
seed = sys.argv[1]
sys.argv = ["fuzz_gpu.py", "--first", seed, "--count", "1"] + sys.argv[2:]
import fuzz_gpu
fuzz_gpu.main() if hasattr(fuzz_gpu, "main") else None
