#!/usr/bin/env python3
"""profiles/r06_summary.json out of the round's committed records (no GPU): the numbers DESIGN.md section 0 quotes, each with the file it is read from."""
import csv
import glob
import json
import os

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def stats(path, needle):
    for row in csv.DictReader(open(os.path.join(P, path))):
        if needle in row["Name"]:
            return {"file": path, "kernel": row["Name"].split("(")[0].replace("void ", ""), "calls": int(row["Calls"]),
                    "avg_us": round(float(row["AverageNs"]) / 1e3, 3), "min_us": round(float(row["MinNs"]) / 1e3, 3), "max_us": round(float(row["MaxNs"]) / 1e3, 3)}
    return None


def line(path):
    d = json.loads(open(os.path.join(P, path)).read().strip().splitlines()[-1])
    r, g = d["roofline"], d["digest"]
    return {"file": path, "bytes": os.path.getsize(os.path.join(P, path)), "value": d["value"], "ms_per_step": d["ms_per_step"],
            "roofline": {k: r.get(k) for k in ("kernel", "kernel_us", "frac", "traffic", "traffic_source", "frac_at_6_sets", "frac_at_1_set", "frac_at_16_sets",
                                                "frac_in16_out1", "frac_in1_out16", "frac_in16_out16")} | {"overlapped": r.get("overlapped"), "copy_ceiling": r.get("copy_ceiling")},
            "parity": d["parity"], "cpu_baseline": {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "omp_value", "omp_cores") if k in d["cpu_baseline"]},
            "digest": {k: g[k] for k in g if k.endswith("frame_ms") or k.endswith("frame_ms_pipelined") or k.endswith("_frac") or k.endswith("kernel_us") or k.endswith("plan_us")}}


out = {"what": "round 6, final library: the evidence behind DESIGN.md section 0", "bench_lines": [line("r06_bench_driver_args.json"), line("r06_bench_plain.json")]}
d = stats("r06_bench_streams1_kernel_stats.csv", "lbs_skin_dyn<true, 7>")
d["frac_of_8TBps"] = round(100e6 / (d["avg_us"] * 1e-6) / 8e12, 4)
c = stats("r06_bench_streams1_kernel_stats.csv", "stream_copy_kernel")
c["frac_of_8TBps"] = round(100e6 / (c["avg_us"] * 1e-6) / 8e12, 4)
out["headline_kernel_alone_under_trace"] = {"lbs_skin_dyn": d, "stream_copy_kernel_same_bytes": c,
                                             "note": "python bench.py --steps 1000 --warmup 100 --opt lbs.streams=1 (LONE launches, 8 rotating sets) under rocprofv3 --kernel-trace --stats"}
d2 = stats("r06_bench_streams2_kernel_stats.csv", "lbs_skin_dyn<true, 7>")
d3 = stats("r06_bench_driver_args_kernel_stats.csv", "lbs_skin_dyn<true, 7>")
out["headline_kernel_overlapped_under_trace"] = {"two_launch_streams": d2, "driver_arguments_whole_run": d3,
                                                  "note": "launches overlapped on two streams: a dispatch lasts longer, two are in flight; `value` is their throughput"}
out["scene_256x1_kernels_under_trace"] = [stats("r06_scene_kernel_stats.csv", n) for n in ("lbs_skin_batch_dyn<true, 7>", "pose_sample_scene_kernel", "pose_update_scene_kernel", "ctrl_copy_kernel")]
out["scene_256x1_batched_kernel_before"] = stats("r06_scene_sampler/scene_kernel_stats_wrap_exits.csv", "lbs_skin_batch<true, 7>")
ex = json.load(open(os.path.join(P, "r06_bench_ex.json")))["results"]
out["lbs_skin_ex_interleaved_output"] = {"file": "r06_bench_ex.json", "us_per_launch": {k: round(v["us_per_launch"], 2) for k, v in ex.items() if isinstance(v, dict) and "us_per_launch" in v},
                                         "before_round_6": {"ex_aos_animated_vertex_68B": 76.87, "ex_aos_static_vertex_48B": 48.41, "ex_4_shapes_aos68": 85.33}}
fz = {"scenarios": 0, "launches": 0, "failures_after_the_fix": 0, "files": 0}
for f in sorted(glob.glob(os.path.join(P, "r06_fuzz", "**", "*.json"), recursive=True)):
    r = json.loads(open(f).read())
    fz["files"] += 1
    fz["scenarios"] += r.get("seeds", 0)
    fz["launches"] += r.get("launches", 0)
    fz["failures_after_the_fix"] += r.get("failures", 0)
out["gpu_fuzz"] = fz | {"found": "root motion over a root node's list with two Positions / Rotations (fixed, seeds pinned); oracle2's by-index handle lookup", "readme": "r06_fuzz/README.md"}
json.dump(out, open(os.path.join(P, "r06_summary.json"), "w"), indent=1)
print(json.dumps(out)[:3000])
