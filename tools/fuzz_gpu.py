#!/usr/bin/env python3
"""Bug hunt on the GPU box: random machines (tests/anim_cases.py::random_machine, plain and listy) far beyond the seeds the
test suite pins, every frame of every scenario against the oracle through tests/test_anim_gpu.py::run_scenario.

    python tools/fuzz_gpu.py --first 100 --count 300 [--listy] [--out gpurun_out/fuzz.json]

Prints one JSON record: seeds run, failures (seed, sampler form, instances, first line of the assertion).  Test infrastructure:
the oracle is the checker here exactly as in tests/."""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--listy", action="store_true")
    ap.add_argument("--bones", type=int, default=7)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import fyrox_amd
    import oracle
    import anim_cases as cases
    import test_anim_gpu as T

    oracle.lib()
    ctx = fyrox_amd.Context(0)
    fails, t0 = [], time.time()
    for seed in range(args.first, args.first + args.count):
        form, n_inst = seed % 3, 1 + seed % 4 if seed % 5 else 70
        sc = cases.random_machine(seed, n_bones=args.bones, listy=args.listy)
        ctx.set_option("anim.sample_form", form)
        o = p = None
        try:
            o, p = T.run_scenario(ctx, oracle, sc, n_instances=n_inst)
            for a in range(len(sc.animations)):
                assert T._drain(lambda: p.pop_event(a, 0)) == T._drain(lambda: o.pop_event(a)), f"events of animation {a}"
        except Exception as e:   # noqa: BLE001 -- every failure is a finding
            fails.append({"seed": seed, "listy": args.listy, "sample_form": form, "instances": n_inst,
                          "error": (str(e).strip().splitlines() or [repr(e)])[0][:300],
                          "where": traceback.format_exc().strip().splitlines()[-3][:200]})
        finally:
            try:
                if p is not None:
                    p.free()
                if o is not None:
                    o.close()
            except Exception:   # noqa: BLE001
                pass
    ctx.set_option("anim.sample_form", 0)
    rec = {"what": "random machines on the GPU against the oracle, every frame", "listy": args.listy, "bones": args.bones,
           "first_seed": args.first, "seeds": args.count, "failures": len(fails), "failed": fails[:40], "seconds": round(time.time() - t0, 1)}
    line = json.dumps(rec)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
