#!/usr/bin/env python3
"""API-sequence fuzz of the host control plane (no GPU): random sequences of builder, setter, query, planning and free calls on a
control-only context, with valid ids and indices mixed with stale, freed, out-of-range and never-created ones.  The contract under test is
the header's: an entry point returns FYX_OK or an error code -- it never aborts, throws across the boundary or touches freed state.  A
sequence FAILS when the process dies (run under tools/asan_control_plane.sh's library for memory errors) or when a call that must work on a
healthy object (a plan of a live animator whose definition was accepted) reports an error other than the ones its arguments earn.

    python tools/fuzz_api_host.py --first 0 --count 300 [--out profiles/r06_fuzz/fuzz_api_host.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fyrox_amd                      # noqa: E402
from fyrox_amd import _native         # noqa: E402
from fyrox_amd import anim as A       # noqa: E402
import anim_cases as cases            # noqa: E402

EARNED = {"FYX_ERR_INVALID_ARG", "FYX_ERR_UNKNOWN_ID", "FYX_ERR_UNSUPPORTED", "FYX_ERR_NO_DEVICE"}


def one_sequence(seed: int) -> dict:
    rng = np.random.default_rng(seed + 77 * 10 ** 6)
    ctx = fyrox_amd.Context(control_only=True)
    lib, h = _native.lib(), ctx._h
    calls, errors = 0, {}
    live = []                          # (animator, scenario)
    dead_ids = [123456789]             # ids that were never created or have been freed

    def attempt(fn, *a, **kw):
        nonlocal calls
        calls += 1
        try:
            return fn(*a, **kw)
        except fyrox_amd.FyxError as e:
            errors[e.status] = errors.get(e.status, 0) + 1
            if e.status not in EARNED:
                raise
            return None

    try:
        for step in range(int(rng.integers(20, 80))):
            r = rng.random()
            if r < 0.18 or not live:                                   # build a product from a random scenario (all builder calls)
                sc = cases.random_machine(int(rng.integers(0, 10 ** 6)), n_bones=int(rng.integers(3, 12)), listy=bool(rng.integers(2)),
                                          lattice=bool(rng.integers(2)))
                p = attempt(cases.build_product, ctx, sc, int(rng.integers(1, 4)))
                if p is not None:
                    live.append((p, sc))
                continue
            p, sc = live[int(rng.integers(0, len(live)))]
            na, nl = len(sc.animations), len(sc.machine.layers)
            bad = lambda n: int(rng.choice([-1, n, n + 7, 2 ** 31 - 1, int(rng.integers(0, max(n, 1)))]))      # mostly out of range
            idx = lambda n: int(rng.integers(0, max(n, 1))) if rng.random() < 0.7 else bad(n)
            if r < 0.40:                                               # plan a frame (the call that must work on a healthy animator)
                mode = 1 if rng.random() < 0.8 else int(rng.integers(-2, 4))
                attempt(p.plan, mode, float(rng.choice([1 / 60, 0.0, -0.1, 1e9, float("nan"), float("inf")])) if rng.random() < 0.2 else 1 / 60)
            elif r < 0.55:                                             # setters with good and bad indices, odd values
                a = idx(na) & 0xffffffff
                inst = int(rng.choice([0, 1, 5, 2 ** 31 - 1, A.ALL_INSTANCES]))
                v = float(rng.choice([0.0, 1.0, -1.0, 1e30, float("nan"), float("inf"), -float("inf")]))
                which = int(rng.integers(0, 7))
                if which == 0: attempt(p.set_speed, a, v, inst)
                elif which == 1: attempt(p.set_time_position, a, v, inst)
                elif which == 2: attempt(p.set_time_slice, a, v, float(rng.choice([v, v + 1, v - 1, float("nan")])), inst)
                elif which == 3: attempt(p.set_loop, a, bool(rng.integers(2)), inst)
                elif which == 4: attempt(p.set_enabled, a, bool(rng.integers(2)), inst)
                elif which == 5: attempt(p.rewind, a, inst)
                else: attempt(p.set_track_enabled, a, idx(40) & 0xffffffff, bool(rng.integers(2)))
            elif r < 0.65:                                             # machine state pokes
                li = idx(nl) & 0xffffffff
                which = int(rng.integers(0, 6))
                if which == 0: attempt(p.layer_state, li, int(rng.choice([0, 1, 9])))
                elif which == 1: attempt(p.set_layer_state, li, idx(4) if rng.random() < 0.8 else -1, idx(4) if rng.random() < 0.5 else -1)
                elif which == 2:
                    kind = int(rng.integers(0, 4))
                    val = {A.PARAM_WEIGHT: float(rng.choice([0.5, float("nan"), -1e30])), A.PARAM_RULE: bool(rng.integers(2)), A.PARAM_INDEX: int(rng.choice([0, 3, 2 ** 31 - 1])),
                           A.PARAM_SAMPLING_POINT: (float(rng.choice([0.5, float("nan")])), 1e38)}[kind]
                    attempt(p.set_parameter, idx(len(sc.machine.parameters)) & 0xffffffff, A.Parameter(kind, val))
                elif which == 3: attempt(p.reset_layer, li)
                elif which == 4: attempt(p.set_transition_state, li, idx(5) & 0xffffffff, float(rng.choice([0.0, 1.0, float("nan")])), float(rng.random()))
                else: attempt(p.set_node_state, li, idx(8) & 0xffffffff, None if rng.random() < 0.5 else idx(4) & 0xffffffff, float(rng.random()))
            elif r < 0.75:                                             # queries
                which = int(rng.integers(0, 6))
                if which == 0: attempt(p.animation_state, idx(na) & 0xffffffff, int(rng.choice([0, 1, 99])))
                elif which == 1: attempt(p.pop_event, idx(na) & 0xffffffff, int(rng.choice([0, 1, 99])))
                elif which == 2: attempt(p.pop_layer_event, idx(nl) & 0xffffffff, int(rng.choice([0, 1, 99])))
                elif which == 3: attempt(p.collect_active_animations_events, idx(nl) & 0xffffffff, int(rng.integers(0, 4)), int(rng.choice([0, 1, 99])))
                elif which == 4: attempt(p.read, int(rng.choice([A.READ_LOCAL_TRS, A.READ_GLOBAL_MATRIX, 77])))      # needs a device: NO_DEVICE
                else: attempt(p.property_slot, idx(12), idx(8))
            elif r < 0.82:                                             # definition edits
                which = int(rng.integers(0, 4))
                if which == 0: attempt(p.remove_animation, idx(na) & 0xffffffff)
                elif which == 1: attempt(p.machine_clear)
                elif which == 2: attempt(p.set_machine, sc.machine)                       # twice: the second must be refused or replace cleanly
                else: attempt(p.add_signal, idx(na) & 0xffffffff, float(rng.random()), bool(rng.integers(2)))
            elif r < 0.90:                                             # frees in the wrong order, of stale ids, twice
                which = int(rng.integers(0, 5))
                calls += 1
                if which == 0: lib.fyx_rig_free(h, ctypes.c_uint64(p.base_id))                # in use by the animator: refused
                elif which == 1: lib.fyx_tracks_data_free(h, ctypes.c_uint64(p.base_id + 1))  # in use by an animation: refused
                elif which == 2: lib.fyx_animator_free(h, ctypes.c_uint64(int(rng.choice(dead_ids))))
                elif which == 3: lib.fyx_rig_free(h, ctypes.c_uint64(int(rng.choice(dead_ids))))
                else: lib.fyx_bone_list_free(h, ctypes.c_uint64(int(rng.choice(dead_ids))))
            elif r < 0.96:                                             # free an animator; its id joins the stale ones; later calls on the wrapper hit UNKNOWN_ID
                k = int(rng.integers(0, len(live)))
                q, _ = live.pop(k)
                attempt(q.free)
                dead_ids.append(q.id)
                attempt(q.plan, 1, 1 / 60)                                                # a call on the freed animator
                calls += 2
                lib.fyx_rig_free(h, ctypes.c_uint64(q.base_id))                            # now nobody uses the rig
                lib.fyx_rig_free(h, ctypes.c_uint64(q.base_id))                            # ... twice
            else:                                                      # scene planning over a random member list incl. stale ids
                ids = [a_.id for a_, _ in live if rng.random() < 0.7] + ([int(rng.choice(dead_ids))] if rng.random() < 0.3 else [])
                arr = np.asarray(ids, np.uint64)
                calls += 1
                lib.fyx_scene_plan(h, arr.ctypes.data_as(ctypes.c_void_p) if len(ids) else None, len(ids), ctypes.c_float(1 / 60))
    finally:
        ctx.close()
    return {"seed": seed, "calls": calls, "errors": errors}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    t0, calls, errors, failed = time.time(), 0, {}, []
    for seed in range(a.first, a.first + a.count):
        try:
            r = one_sequence(seed)
            calls += r["calls"]
            for k, v in r["errors"].items():
                errors[k] = errors.get(k, 0) + v
        except Exception as e:     # noqa: BLE001
            failed.append({"seed": seed, "what": repr(e)[:300]})
            print(json.dumps(failed[-1]), flush=True)
    rec = {"what": "random API sequences on a control-only context (no GPU): calls return codes, the process survives", "first_seed": a.first, "sequences": a.count,
           "calls": calls, "error_codes_returned": errors, "failures": len(failed), "failed": failed, "seconds": round(time.time() - t0, 1)}
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
