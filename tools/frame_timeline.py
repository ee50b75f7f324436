#!/usr/bin/env python3
"""Timeline of a repeating frame out of a rocprofv3 kernel trace (`*_kernel_trace.csv`): for the most frequent run of
consecutive kernels ending in the anchor kernel, the median duration of each kernel and the median idle gap ahead of it.

    python tools/frame_timeline.py TRACE.csv [--anchor lbs_skin] [--last N]

A frame of one character is a chain of dependent launches on one stream; what the chain costs beyond the kernels' own
durations is the gaps -- this tool puts a number on each."""
import argparse
import collections
import csv
import json
import re
import statistics


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("fyx::", "").replace("void ", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--anchor", default="lbs_skin")
    ap.add_argument("--last", type=int, default=0, help="use only the last N dispatches of the trace")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
                         int(r["Workgroup_Size_X"])))
    rows.sort()
    if a.last:
        rows = rows[-a.last:]
    # frames = runs of dispatches that end with the anchor kernel
    frames, cur = [], []
    for r in rows:
        cur.append(r)
        if a.anchor in r[2]:
            frames.append(cur)
            cur = []
    shapes = collections.Counter(tuple((k[2], k[3], k[4]) for k in f) for f in frames)
    out = []
    for shape, n in shapes.most_common(4):
        if n < 8:
            continue
        fs = [f for f in frames if tuple((k[2], k[3], k[4]) for k in f) == shape]
        rec = {"frames": n, "kernels": []}
        for i, (name, grid, wg) in enumerate(shape):
            dur = statistics.median(f[i][1] - f[i][0] for f in fs) / 1e3
            gap = statistics.median(f[i][0] - f[i - 1][1] for f in fs) / 1e3 if i else None
            rec["kernels"].append({"kernel": name, "threads": grid, "block": wg, "us": round(dur, 2), "gap_before_us": None if gap is None else round(gap, 2)})
        rec["first_start_to_last_end_us"] = round(statistics.median(f[-1][1] - f[0][0] for f in fs) / 1e3, 2)
        rec["sum_of_kernels_us"] = round(sum(k["us"] for k in rec["kernels"]), 2)
        out.append(rec)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
