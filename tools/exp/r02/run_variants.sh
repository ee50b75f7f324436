#!/bin/bash
# GPU box: time the listed library variants (tools/exp/libs) on the lone-launch bench.  Usage: run_variants.sh ONLY TAG...
only=$1; shift
mkdir -p gpurun_out/var
python tools/tune_dyn.py --only "$only" --tag base > gpurun_out/var/base.json 2> gpurun_out/var/base.err
for tag in "$@"; do
  FYX_LIB_PATH=$PWD/tools/exp/libs/libfyrox_hip_$tag.so python tools/tune_dyn.py --only "$only" --tag $tag > gpurun_out/var/$tag.json 2> gpurun_out/var/$tag.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/var/*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(d["tag"], "mismatching:", d["mismatching"])
    for r in d["rows"]:
        print("   s%d %-24s %6.2f us (min %6.2f)" % (r["streams"], r["variant"], r["us_median"], r["us_min"]))
PY
