#!/usr/bin/env python3
"""Condense gpurun_out/prof (written by tools/profile.sh on the GPU box) into the tracked
profiles/ directory: kernel-trace stats, PMC medians with the gfx950 FETCH_SIZE correction, and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic)."""
import collections, csv, json, os, shutil, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
dst = "profiles"
os.makedirs(dst, exist_ok=True)

for sub, name in (("trace", f"{tag}_bench_streams2_kernel_stats.csv"), ("trace_s1", f"{tag}_bench_streams1_kernel_stats.csv")):
    p = os.path.join(src, sub, "bench_kernel_stats.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
p = os.path.join(src, "trace_pose", "pose_kernel_stats.csv")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, f"{tag}_crowd_pose_kernel_stats.csv"))
p = os.path.join(src, "trace_scene", "scene_kernel_stats.csv")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, f"{tag}_scene_kernel_stats.csv"))
p = os.path.join(src, "trace_ex", "ex_kernel_stats.csv")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, f"{tag}_bench_ex_kernel_stats.csv"))
for f in ("bench_driver_args.json", "bench_plain.json", "bench_under_trace.json", "bench_under_trace_s1.json", "pose_plain.json", "pose_under_trace.json",
          "pose_root_motion.json", "pose_fused.json", "pose_palette_output.json", "bench_ex.json", "bench_ex_under_trace.json", "timeline.json",
          "write_ceiling.json", "calibration_stream.json", "scene_64x4.json", "scene_256x1.json", "scene_under_trace.json",
          "character_plain.json", "character_under_trace.json", "character_timeline.json", "update_stamps.json"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{f}"))


for f in ("crowd_lone_under_trace.jsonl", "crowd_lone.jsonl"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{f}"))
p = os.path.join(src, "trace_character", "chr_kernel_stats.csv")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, f"{tag}_character_kernel_stats.csv"))
p = os.path.join(src, "trace_crowd_lone", "crowd_kernel_stats.csv")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, f"{tag}_crowd_lone_kernel_stats.csv"))


def crowd_spread():
    """Why lbs_skin_crowd's duration spreads in the C3 frame trace (tools/bench_pose.py): every dispatch of the per-dispatch trace
    with what ran beside it -- the upload stream's copy of the NEXT frame's control block and nothing else on one launch stream --
    split into the frame loops (pose kernels before and after it on the same stream) and the skinning-only loop."""
    p = os.path.join(src, "trace_pose", "pose_kernel_trace.csv")
    if not os.path.exists(p):
        return None
    rows = list(csv.DictReader(open(p)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:28]) for r in rows)
    crowd = [(s, e) for s, e, n in ev if n.startswith("lbs_skin_crowd")]
    others = [(s, e, n) for s, e, n in ev if not n.startswith("lbs_skin_crowd")]
    out = {"dispatches": len(crowd)}
    groups = {"alone": [], "beside_another_kernel": []}
    j = 0
    for s, e in crowd:
        beside = [n for os_, oe, n in others if os_ < e and oe > s]
        groups["beside_another_kernel" if beside else "alone"].append((e - s) / 1e3)
    for k, v in groups.items():
        if v:
            v.sort()
            out[k] = {"n": len(v), "min_us": v[0], "median_us": v[len(v) // 2], "max_us": v[-1], "mean_us": sum(v) / len(v)}
    # previous kernel on the timeline: the first crowd dispatch after a pose kernel vs after another crowd dispatch
    prev = {"after_a_pose_kernel": [], "after_a_crowd_dispatch": []}
    names = [(s, e, n) for s, e, n in ev]
    for i, (s, e, n) in enumerate(names):
        if not n.startswith("lbs_skin_crowd") or i == 0:
            continue
        pn = names[i - 1][2]
        prev["after_a_crowd_dispatch" if pn.startswith("lbs_skin_crowd") else "after_a_pose_kernel"].append((e - s) / 1e3)
    for k, v in prev.items():
        if v:
            v.sort()
            out[k] = {"n": len(v), "min_us": v[0], "median_us": v[len(v) // 2], "max_us": v[-1], "mean_us": sum(v) / len(v)}
    return out


def trace_summary(sub):
    """per-launch throughput time from the kernel trace (first start -> last end over N launches)"""
    p = os.path.join(src, sub, "bench_kernel_trace.csv")
    if not os.path.exists(p):
        return None
    rows = list(csv.DictReader(open(p)))
    dyn = [r for r in rows if "lbs_skin_dyn<" in r["Kernel_Name"]]      # the headline kernel of the C4 launch (round 2 on)
    rows = dyn if dyn else [r for r in rows if "lbs_skin<" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # 1000 consecutive launches out of the first timed region (after the warm-up and the pilot pass); the bench's
    # later legs (serialized kernel time, the other BASELINE configs) launch other things
    rows = rows[1200:2200] if len(rows) >= 2200 else rows[-1000:]
    st = [int(r["Start_Timestamp"]) for r in rows]
    en = [int(r["End_Timestamp"]) for r in rows]
    dur = sorted(e - s for s, e in zip(st, en))
    return {"launches": len(rows), "span_us_per_launch": (max(en) - min(st)) / len(rows) / 1e3,
            "kernel_duration_us_avg": sum(dur) / len(dur) / 1e3, "kernel_duration_us_median": dur[len(dur) // 2] / 1e3}


pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_lds"):
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        k = ("lbs_skin_aos" if "lbs_skin_aos<" in n else "lbs_skin_ex" if "lbs_skin_ex<" in n else
             "lbs_skin_crowd" if "lbs_skin_crowd<" in n else "lbs_skin_batch" if "lbs_skin_batch<" in n else
             "lbs_skin" if ("lbs_skin<" in n or "lbs_skin_dyn<" in n) else
             "stream_copy" if "stream_copy" in n else None)
        if k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        v.sort()
        pmc.setdefault(k, {})[c] = {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "launches": len(v)}

out = {"source": src, "trace_streams2": trace_summary("trace"), "trace_streams1": trace_summary("trace_s1"), "pmc": pmc,
       "crowd_duration_spread_in_the_c3_frame_trace": crowd_spread()}
if "stream_copy" in pmc and "FETCH_SIZE" in pmc["stream_copy"] and "lbs_skin" in pmc:
    KB = 1024.0
    copy_rd_known, copy_wr_known = 48 * 1_250_000, 32 * 1_250_000
    f_rd = copy_rd_known / (pmc["stream_copy"]["FETCH_SIZE"]["median"] * KB)   # ~2.0 on gfx950 (guide: FETCH_SIZE counts 1/2)
    f_wr = copy_wr_known / (pmc["stream_copy"]["WRITE_SIZE"]["median"] * KB)   # ~1.0
    rd = pmc["lbs_skin"]["FETCH_SIZE"]["median"] * KB * f_rd
    wr = pmc["lbs_skin"]["WRITE_SIZE"]["median"] * KB * f_wr
    out["calibration"] = {"known_copy_read_bytes": copy_rd_known, "known_copy_write_bytes": copy_wr_known,
                          "fetch_size_correction": f_rd, "write_size_correction": f_wr}
    out["lbs_hbm_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr, "algorithmic": 100_000_000}
    algo = {"lbs_skin_ex": ("4 blend shapes -> SoA", 172_000_000), "lbs_skin_aos": ("vertex buffer in -> out, 68 B", 136_000_000),
            "lbs_skin_crowd": ("100 instances x 10 k vertices / 64 bones: unique bytes", 600_000 + 100 * 4096 + 40_000_000),
            "lbs_skin_batch": ("64 meshes x 20 k vertices / 64 bones in one launch", 64 * (20_000 * 100 + 4096))}
    for k, (what, ab) in algo.items():
        if k in pmc and "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]:
            r_, w_ = pmc[k]["FETCH_SIZE"]["median"] * KB * f_rd, pmc[k]["WRITE_SIZE"]["median"] * KB * f_wr
            out[k + "_hbm_bytes_per_launch"] = {"what": what, "read": r_, "write": w_, "total": r_ + w_, "algorithmic": ab}
    json.dump({"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile.sh), KB units, corrected by "
                         "the factors measured on fyx_calib_stream_copy's known 60 MB read / 40 MB written in the same run "
                         f"(FETCH x{f_rd:.3f}, WRITE x{f_wr:.3f}); see profiles/{tag}_summary.json"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
# scene pose kernels (fyx_scene_update over 256 characters): HBM bytes per launch, same corrections
scene = {}
for sub, cname in (("pmc_scene_fetch", "FETCH_SIZE"), ("pmc_scene_write", "WRITE_SIZE")):
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] != cname or "scene_kernel" not in n and "lbs_skin_batch<" not in n:
            continue
        short = n.split("(")[0].split("::")[-1].split("<")[0]
        agg[short].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v.sort()
        scene.setdefault(k, {})[cname] = {"median_KB": v[len(v) // 2], "launches": len(v)}
if scene and "calibration" in out:
    for k, d in scene.items():
        rd = d.get("FETCH_SIZE", {}).get("median_KB", 0.0) * 1024.0 * out["calibration"]["fetch_size_correction"]
        wr = d.get("WRITE_SIZE", {}).get("median_KB", 0.0) * 1024.0 * out["calibration"]["write_size_correction"]
        d["hbm_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr}
    out["scene_256x1x5k"] = {"what": "tools/bench_scene.py --characters 256 --instances 1 --verts 5000 (batched): one launch per stage per frame", "kernels": scene}
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
