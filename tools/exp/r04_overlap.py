#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace of the pipelined C3 frame (tools/exp/r04_frame.py WHICH=trace) and says how much of the pose
kernels' time lies INSIDE a skinning dispatch of another frame: per frame (= one lbs_skin_crowd dispatch) the period between
consecutive skinning starts, the skinning duration, and for the pose kernels that started during it their own durations."""
import csv, json, statistics, sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        kind = "skin" if "lbs_skin_crowd" in n else "sample" if "pose_sample" in n else "update" if "pose_update" in n else "copy" if "ctrl_copy" in n or "copyBuffer" in n else None
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
last = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rows = rows[-last:]
skins = [r for r in rows if r[2] == "skin"]
out = {"skin_dispatches": len(skins)}
if len(skins) > 3:
    out["skin_start_to_start_us"] = round(statistics.median(b[0] - a[0] for a, b in zip(skins, skins[1:])) / 1e3, 2)
    out["skin_us"] = round(statistics.median(s[1] - s[0] for s in skins) / 1e3, 2)
    out["gap_between_skins_us"] = round(statistics.median(b[0] - a[1] for a, b in zip(skins, skins[1:])) / 1e3, 2)
    for kind in ("sample", "update", "copy"):
        ks = [r for r in rows if r[2] == kind]
        if not ks:
            continue
        inside = [k for k in ks if any(s[0] <= k[0] and k[1] <= s[1] for s in skins)]
        started_inside = [k for k in ks if any(s[0] <= k[0] < s[1] for s in skins)]
        out[kind] = {"dispatches": len(ks), "median_us": round(statistics.median(k[1] - k[0] for k in ks) / 1e3, 2),
                     "entirely_inside_a_skin": len(inside), "started_inside_a_skin": len(started_inside)}
print(json.dumps(out))
