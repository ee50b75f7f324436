#!/usr/bin/env python3
"""Condense gpurun_out/prof05 (tools/profile_r05.sh) into the tracked profiles/r05_* files and profiles/hbm_traffic.json."""
import collections, csv, glob, json, os, shutil, statistics, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof05"
dst, tag = "profiles", "r05"


def last_json(path):
    try:
        lines = [l for l in open(path) if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def stats_row(path, needle):
    try:
        for r in csv.DictReader(open(path)):
            if needle in r["Name"]:
                return {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                        "stddev_us": float(r["StdDev"]) / 1e3}
    except OSError:
        pass
    return None


def find(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None


summary = {}
# 1. the headline in three traced processes + one untraced: kernel-trace average against the same process's per-dispatch events
procs = []
for k in (1, 2, 3):
    line = last_json(os.path.join(src, f"head{k}.json"))
    st = stats_row(find(f"head{k}/**/bench_kernel_stats.csv") or "", "lbs_skin_dyn")
    if line and st:
        ev = line["roofline"]
        procs.append({"process": k, "under_trace": True, "trace_avg_us": st["avg_us"], "trace_calls": st["calls"], "trace_min_us": st["min_us"], "trace_max_us": st["max_us"],
                      "events_kernel_us_median_of_allocations": ev["kernel_us"], "events_per_allocation": ev.get("kernel_us_per_allocation"),
                      "events_over_trace": ev["kernel_us"] / st["avg_us"], "frac_by_trace": 100e6 / (st["avg_us"] * 1e-6) / 8e12,
                      "frac_by_events": ev["frac"], "copy_ceiling_frac": (ev.get("copy_ceiling") or {}).get("frac")})
line = last_json(os.path.join(src, "head_untraced.json"))
if line:
    ev = line["roofline"]
    procs.append({"process": "untraced", "under_trace": False, "events_kernel_us_median_of_allocations": ev["kernel_us"], "events_per_allocation": ev.get("kernel_us_per_allocation"),
                  "frac_by_events": ev["frac"], "copy_ceiling_frac": (ev.get("copy_ceiling") or {}).get("frac"), "value": line["value"]})
json.dump({"what": "the headline leg (C4, 1 M vertices / 256 bones, eight rotating sets, ONE launch stream) in separate processes of one gpurun call: rocprofv3 --kernel-trace "
                   "--stats average of lbs_skin_dyn over every dispatch of the process, beside the same process's own per-dispatch events (hipExtLaunchKernel start / stop: "
                   "bench.py's roofline.kernel_us, the median over four allocations of the output sets, each the average of 1000 launches)",
           "processes": procs}, open(os.path.join(dst, f"{tag}_headline_processes.json"), "w"), indent=1)
summary["headline_processes"] = procs
for sub, name in (("head1", f"{tag}_bench_streams1_kernel_stats.csv"), ("streams2", f"{tag}_bench_streams2_kernel_stats.csv"), ("crowd_lone", f"{tag}_crowd_lone_kernel_stats.csv"),
                  ("pose", f"{tag}_crowd_pose_kernel_stats.csv"), ("character", f"{tag}_character_kernel_stats.csv"), ("scene", f"{tag}_scene_kernel_stats.csv"),
                  ("frame_skin", f"{tag}_frame_skin_kernel_stats.csv")):
    p = find(f"{sub}/**/*_kernel_stats.csv")
    if p:
        shutil.copy(p, os.path.join(dst, name))
for f in ("bench_plain.json", "bench_driver_args.json", "bench_2ranks_one_gpu_test_hook.json", "bench_one_process_one_gpu_test_hook.json", "frame_skin.jsonl",
          "frame_skin_under_trace.jsonl", "scene_records.jsonl", "vertex_buffer.jsonl", "crowd_lone.jsonl", "crowd_lone_under_trace.jsonl", "c3_timeline.jsonl",
          "character_plain.json", "character_under_trace.json", "scene_under_trace.json", "pose_under_trace.json", "streams2.json"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{f}"))
st2 = stats_row(find("streams2/**/bench_kernel_stats.csv") or "", "lbs_skin_dyn")
if st2:
    summary["trace_streams2_lbs_skin_dyn"] = st2

# 2. counters: groups of N dispatches in launch order (tools/pmc_probe_r04.py)
GROUPS = ["stream_copy (60 MB read + 40 MB written)", "C4 coherent bone indices", "C4 random bone indices", "C3 crowd coherent", "C3 crowd random"]


def pmc(dirname):
    p = find(f"{dirname}/**/pmc_counter_collection.csv")
    if not p:
        return None
    per = collections.defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if not any(s in n for s in ("stream_copy", "lbs_skin")):
            continue
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        names[int(r["Dispatch_Id"])] = n.split("(")[0].split("::")[-1][:40]
    ids = sorted(per)
    # cut where the kernel name or the run changes: copy | dyn | dyn | crowd | crowd, equal counts within a name
    runs, cur = [], []
    for i in ids:
        if cur and names[i] != names[cur[-1]]:
            runs.append(cur)
            cur = []
        cur.append(i)
    if cur:
        runs.append(cur)
    groups = []
    for run in runs:
        if "stream_copy" in names[run[0]]:
            groups.append(run)
        else:      # coherent then random: two halves
            h = len(run) // 2
            groups += [run[:h], run[h:]]
    out = {}
    for g, label in zip(groups, GROUPS):
        cs = collections.defaultdict(list)
        for i in g[2:]:             # the first launches of a group warm the caches
            for c, v in per[i].items():
                cs[c].append(v)
        out[label] = {"kernel": names[g[0]], "launches": len(g) - 2, **{c: statistics.median(v) for c, v in cs.items()}}
    return out


fetch, write, lds = pmc("pmc_FETCH_SIZE"), pmc("pmc_WRITE_SIZE"), pmc("pmc_SQ_LDS_BANK_CONFLICT")
hbm = {}
if fetch and write:
    # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md):
    # calibrate both on the stream copy of known bytes, then apply the factors
    cal = GROUPS[0]
    f_factor = 60e6 / (fetch[cal]["FETCH_SIZE"] * 1024) if fetch[cal].get("FETCH_SIZE") else None
    w_factor = 40e6 / (write[cal]["WRITE_SIZE"] * 1024) if write[cal].get("WRITE_SIZE") else None
    for g in GROUPS[1:]:
        if g in fetch and g in write and f_factor and w_factor:
            rd, wr = fetch[g]["FETCH_SIZE"] * 1024 * f_factor, write[g]["WRITE_SIZE"] * 1024 * w_factor
            alg = 100e6 if g.startswith("C4") else 404.696e6
            hbm[g] = {"read_bytes": rd, "written_bytes": wr, "total_bytes": rd + wr, "algorithmic_bytes": alg, "ratio": (rd + wr) / alg}
    summary["hbm_calibration"] = {"fetch_factor_on_stream_copy": f_factor, "write_factor_on_stream_copy": w_factor,
                                  "note": "factor = known bytes of the stream copy / counter bytes: FETCH_SIZE counts 128-byte requests as 64 on gfx950 (factor ~2), WRITE_SIZE ~1"}
    summary["hbm_bytes_per_launch"] = hbm
    if GROUPS[1] in hbm:
        json.dump({"hbm_bytes_per_launch": hbm[GROUPS[1]]["total_bytes"], "read": hbm[GROUPS[1]]["read_bytes"], "written": hbm[GROUPS[1]]["written_bytes"],
                   "algorithmic": 100e6, "source": "tools/profile_r05.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/pmc_probe_r04.py, calibrated on the "
                   "stream copy of known bytes in the same pass", "round": 5}, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
if lds:
    summary["lds_counters_per_launch"] = {g: {k: v for k, v in d.items()} for g, d in lds.items() if g != GROUPS[0]}
# 3. the lines
for key, f in (("bench_plain", "bench_plain.json"), ("bench_driver_args", "bench_driver_args.json")):
    d = last_json(os.path.join(src, f))
    if not d:
        continue
    e = d.get("extra", {})
    pick = lambda r, ks: {k: r[k] for k in ks if isinstance(r, dict) and k in r}
    summary[key] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "roofline": pick(d["roofline"], ("kernel_us", "kernel_us_min", "kernel_us_max", "frac", "traffic")),
                    "overlapped_frac": d["roofline"]["overlapped"]["frac"], "copy_ceiling_frac": (d["roofline"].get("copy_ceiling") or {}).get("frac"),
                    **{k: pick(e.get(k, {}), ("frame_ms", "frame_mode", "frame_ms_pipelined", "frame_ms_one_stream", "frame_ms_one_launch", "frame_ms_skin_outputs", "pose_ms", "skin_ms", "frame_over_skin", "frame_roofline_frac",
                                               "kernel_us", "frac", "slowdown_vs_coherent", "error")) for k in ("c2", "c3", "c3_root_motion", "c5", "scene_64x4", "scene_256x1",
                                                                                                                 "c4_random_bones", "c3_random_bones")},
                    "c3_kernel": pick(e.get("c3", {}).get("roofline", {}), ("kernel_us", "frac", "kernel_us_in_frame", "frac_in_frame")),
                    "c3_fused_kernel": pick(e.get("c3_fused", {}).get("roofline", {}), ("kernel_us", "frac")),
                    "cpu_baseline": pick(d.get("cpu_baseline", {}), ("value", "cores", "omp_value", "omp_cores"))}
d = last_json(os.path.join(src, "bench_2ranks_one_gpu_test_hook.json"))
if d:
    summary["two_ranks_on_one_gpu_test_hook"] = {k: d.get(k) for k in ("n_gpus", "value", "strong_value", "strong_with_gather_value", "strong_with_gather_form", "crowd_value", "crowd_frame_ms")}
    summary["two_ranks_on_one_gpu_test_hook"]["comm_error"] = d.get("extra", {}).get("strong_scaling", {}).get("comm_error")
for name, needle in (("crowd_lone_exact", "lbs_skin_crowd<512, true"), ("crowd_lone_fused", "lbs_skin_crowd<512, false")):
    st = stats_row(os.path.join(dst, f"{tag}_crowd_lone_kernel_stats.csv"), needle)
    if st:
        summary[name + "_trace"] = {**st, "frac": 404.696e6 / (st["avg_us"] * 1e-6) / 8e12}
for name, needle in (("pose_sample_crowd", "pose_sample_crowd"), ("pose_update", "pose_update_kernel"), ("ctrl_copy", "ctrl_copy"), ("crowd_in_frame", "lbs_skin_crowd")):
    st = stats_row(os.path.join(dst, f"{tag}_crowd_pose_kernel_stats.csv"), needle)
    if st:
        summary.setdefault("c3_frame_kernels_one_chain_trace", {})[name] = st
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
