#!/usr/bin/env python3
"""bench.py -- skinned vertices/s of the HIP linear-blend-skinning hot path on MI355X.

A "step" = one pass of the hot path over one batch: ONE launch of the skinning kernel over the C4 workload
(1 M vertices / 256 bones; position + normal + tangent, 4 influences) through the C ABI (fyx_lbs_skin_device),
inputs resident in HBM.  Steps rotate through `--sets` disjoint buffer sets (default 8 x 100 MB > 2 x the 256 MiB
Infinity Cache) so the stream comes from HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

What the one JSON line says (rank 0 prints it):
  value / ms_per_step   steps timed between barrier + device sync on both sides, max over ranks.  K = 20 steps of a 16 us
                        kernel last 0.3 ms, of which opening and closing the region is ~15 %, so a timed region is `repeats`
                        back-to-back passes of K steps (no sync inside; `timed_steps` = K x repeats >= 20 ms) and the median
                        of five regions is reported: the per-step number does not depend on K.
  roofline.frac         follows from ONE kernel: algorithmic bytes / the kernel's own duration (launches serialized on
                        one stream, HIP events: what rocprofv3 --kernel-trace reports per dispatch).
  roofline.overlapped   the same bytes / the time per launch of the timed region, where independent launches overlap on
                        the library's launch streams (head of one launch under the tail of the previous one).
  roofline.copy_ceiling a no-math 60 MB-in / 40 MB-out copy kernel on the same buffers' sizes, same run, same box.
  extra.c2 / c3 / c5    the other BASELINE configs end to end (pose -> palette -> skinning), N = 1 only.
  extra.scene_*         the scene tick: many distinct characters per frame, one fyx_scene_update + one fyx_lbs_skin_batch.
  extra.c3_fused        C3's skinning launch with lbs.exact = 0 (inside north_star's 1e-5), with its measured max_rel_err.
  extra.vertex_buffer   68-byte AnimatedVertex in, render-ready vertex buffer out (lbs_skin_aos), with and without 4 blend shapes.
  extra.strong_scaling  N > 1: the SAME 1 M-vertex mesh cut by vertex range over the N GPUs (BASELINE config 4), compute
                        only and with the RCCL exchange (fyx_allgather_skinned) in BOTH of its forms (comm.form 0 / 1).
  extra.crowd_scaling   N > 1: C3 cut by instance range, per-rank pose + skinning, no communication.
  N > 1                 `value` = BASELINE config 4 as written (strong scaling, compute only; = strong_value); value_with_gather = the same
                        with the fastest exchange form; weak_value = every GPU its own 1 M vertices, no collective; crowd_value = C3 by instances.
The line is a DIGEST (< 6 KB: contract keys, roofline, parity, cpu_baseline, numbers of the sub-records); the full record with every
sub-record, note and box fact goes to bench_full.json (`full_record` in the line names it).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from functools import partial

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_VERTS = 1_000_000
N_BONES = 256
BYTES_PER_VERTEX = 100          # 60 read + 40 written
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MIN_REGION_MS = 20.0            # a timed region shorter than this is repeated (see module docstring)
CHECK_THREADS = 8               # OpenMP threads of the oracle in the parity legs.  NOT all cores: on hosts with a CPU quota a burst of
                                # 128 threads right before a host-bound timed loop gets the process throttled, and the loop measures that
EXCHANGE_TIMEOUT_S = 120        # N > 1: the RCCL exchange leg may take this long before the line is printed without it


def _read(path: str):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:     # noqa: BLE001
        return None


def box_facts(device_index: int = 0) -> dict:
    """What tells a fast box from a slow one, read where the driver publishes it (sysfs of the amdgpu card, hwmon; rocm-smi as a
    second source): current / available sclk and mclk, power now and cap, temperatures, compute- and memory-partition modes, driver
    and firmware versions.  Never fails: a missing file is a missing key.  Called at the start, right behind the headline's timed
    regions and at the end of the run, so that a clock that dropped under load shows."""
    import glob
    import shutil
    import subprocess
    out = {"t": time.time()}
    try:
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            if _read(os.path.join(d, "vendor")) == "0x1002" and os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                cards.append(d)
        out["amdgpu_cards"] = len(cards)
        if cards:
            d = cards[min(device_index, len(cards) - 1)]
            out["card"] = os.path.basename(os.path.dirname(d))

            def dpm(name):
                txt = _read(os.path.join(d, name))
                if not txt:
                    return None
                levels = [ln.strip() for ln in txt.splitlines() if ln.strip()]
                cur = [ln for ln in levels if ln.endswith("*")]
                return {"current": cur[0].rstrip("*").strip() if cur else None, "levels": levels}
            for key, name in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk"), ("socclk", "pp_dpm_socclk")):
                v = dpm(name)
                if v is not None:
                    out[key] = v
            for key in ("current_compute_partition", "current_memory_partition", "available_compute_partition", "available_memory_partition",
                        "power_dpm_force_performance_level", "gpu_busy_percent", "mem_busy_percent", "vbios_version", "unique_id", "device", "revision"):
                v = _read(os.path.join(d, key))
                if v is not None:
                    out[key] = v
            for hm in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                for key in ("power1_cap", "power1_cap_max", "power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input"):
                    v = _read(os.path.join(hm, key))
                    if v is not None:
                        try:
                            out["hwmon_" + key] = int(v)
                        except ValueError:
                            out["hwmon_" + key] = v
        out["amdgpu_driver_version"] = _read("/sys/module/amdgpu/version")
        out["kernel"] = _read("/proc/sys/kernel/osrelease")
        out["rocm_version"] = _read("/opt/rocm/.info/version")
        smi = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        if smi:
            try:
                cp = subprocess.run([smi, "-d", str(device_index), "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--showmemorypartition",
                                     "--showcomputepartition", "--showdriverversion", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
                txt = cp.stdout.strip()
                start = txt.find("{")
                out["rocm_smi"] = json.loads(txt[start:]) if start >= 0 else {"rc": cp.returncode, "stderr": cp.stderr[-300:]}
            except Exception as e:     # noqa: BLE001
                out["rocm_smi"] = {"error": repr(e)}
        else:
            out["rocm_smi"] = None
    except Exception as e:     # noqa: BLE001
        out["error"] = repr(e)
    return out


def pmc_traffic(n_verts: int, n_bones: int, timeout_s: float = 150.0) -> dict:
    """roofline.traffic measured in THIS run: two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE, each with --kernel-trace only, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: separate passes, no other trace domain) over tools/pmc_probe_headline.py -- a dozen
    launches of a stream copy of KNOWN bytes (60 MB read + 40 MB written) and a dozen of the headline kernel.  The counters are in KB and
    FETCH_SIZE counts a 128-byte request as 64 on gfx950: both are calibrated on the copy in the same pass.  Returns {"hbm_bytes_per_launch",
    "read", "written", factors, "source"} or {"error": ...}; never raises."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof:
        return {"error": "rocprofv3 is not installed"}
    probe = os.path.join(ROOT, "tools", "pmc_probe_headline.py")
    out = {}
    tmp = tempfile.mkdtemp(prefix="fyx_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cp = subprocess.run([prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, probe, "12",
                                 str(n_verts), str(n_bones)], cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if cp.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {counter}: rc {cp.returncode}, {'no' if not files else 'a'} counter file; {cp.stderr[-300:]!r}"}
            groups = {"copy": [], "skin": []}
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] != counter:
                    continue
                name = r["Kernel_Name"]
                key = "copy" if "stream_copy" in name else "skin" if "lbs_skin" in name else None
                if key:
                    groups[key].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            if len(groups["copy"]) < 4 or len(groups["skin"]) < 4:
                return {"error": f"--pmc {counter}: {len(groups['copy'])} copy and {len(groups['skin'])} skinning dispatches in the counter file"}
            med = {k: statistics.median(v for _, v in sorted(g)[2:]) for k, g in groups.items()}     # the first launches warm the caches
            known = 60e6 if counter == "FETCH_SIZE" else 40e6
            factor = known / (med["copy"] * 1024.0)
            out[counter] = {"factor_on_the_stream_copy": factor, "bytes": med["skin"] * 1024.0 * factor, "raw_kb": med["skin"], "dispatches": len(groups["skin"]) - 2}
        return {"hbm_bytes_per_launch": out["FETCH_SIZE"]["bytes"] + out["WRITE_SIZE"]["bytes"], "read": out["FETCH_SIZE"]["bytes"], "written": out["WRITE_SIZE"]["bytes"],
                "fetch_factor_on_stream_copy": out["FETCH_SIZE"]["factor_on_the_stream_copy"], "write_factor_on_stream_copy": out["WRITE_SIZE"]["factor_on_the_stream_copy"],
                "dispatches_per_counter": out["FETCH_SIZE"]["dispatches"],
                "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/pmc_probe_headline.py, "
                          "calibrated on the stream copy of known bytes in the same pass (FETCH_SIZE counts 128-byte requests as 64 on gfx950)"}
    except subprocess.TimeoutExpired:
        return {"error": f"a rocprofv3 pass did not finish within {timeout_s:.0f} s"}
    except Exception as e:     # noqa: BLE001
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def emit(line: str) -> None:
    """Rank 0's ONE JSON line, as the last thing on stdout: whatever native libraries have buffered in C stdio (RCCL prints
    a version banner through it, which would otherwise come out at exit, after the line) is flushed first."""
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:     # noqa: BLE001
        pass
    print(line, flush=True)


LINE_LIMIT = 6144              # bytes: the driver's parser gave up on round 5's 34 KB line


def _dig(d, *path):
    """d[path[0]][path[1]]... or None: a sub-record that did not run is a missing number, never an exception."""
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _num(x, digits=5):
    """Numbers of the compact line: 5 significant digits (the full record keeps what was measured)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    return x


def compact_record(full: dict, full_path: str | None) -> dict:
    """The ONE line the driver parses: the contract's keys, `roofline`, `parity`, `cpu_baseline`, and a DIGEST of the sub-records --
    numbers only, no prose.  Everything else (box facts at three points of the run, per-set sweeps, notes, the sub-records whole) is
    the full record, written beside it (`full_record`)."""
    r = full.get("roofline") or {}
    ex = full.get("extra") or {}
    cfg = full.get("config") or {}
    roof = None
    if r:
        tsrc = r.get("traffic_source") or ""
        roof = {"bound": r.get("bound"), "kernel": r.get("kernel"), "kernel_us": _num(r.get("kernel_us")),
                "achieved": _num(r.get("achieved")), "peak": r.get("peak"), "unit": r.get("unit"), "frac": _num(r.get("frac")),
                "traffic": _num(r.get("traffic"), 7),
                "traffic_source": None if r.get("traffic") is None else "pmc_in_this_run" if tsrc.startswith("measured in this run") else "replayed_builder_pmc",
                "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"),
                "sets": cfg.get("sets"),
                "frac_at_6_sets": _num(r.get("frac_at_6_sets")), "kernel_us_at_6_sets": _num(r.get("kernel_us_at_6_sets")),
                "frac_at_1_set": _num(_dig(r, "by_number_of_rotating_sets", "1", "frac")),
                "frac_at_16_sets": _num(_dig(r, "by_number_of_rotating_sets", "16", "frac")),
                "frac_in16_out1": _num(_dig(r, "by_number_of_rotating_sets", "inputs_vs_outputs", "in16_out1", "frac")),
                "frac_in16_out6": _num(_dig(r, "by_number_of_rotating_sets", "inputs_vs_outputs", "in16_out6", "frac")),
                "frac_in16_out16": _num(_dig(r, "by_number_of_rotating_sets", "inputs_vs_outputs", "in16_out16", "frac")),
                "frac_in1_out16": _num(_dig(r, "by_number_of_rotating_sets", "inputs_vs_outputs", "in1_out16", "frac")),
                "kernel_us_min": _num(r.get("kernel_us_min")), "kernel_us_max": _num(r.get("kernel_us_max")),
                "overlapped": {"frac": _num(_dig(r, "overlapped", "frac")), "avg_launch_us": _num(_dig(r, "overlapped", "avg_launch_us")),
                               "launch_streams": _dig(r, "overlapped", "launch_streams")},
                "copy_ceiling": {"frac": _num(_dig(r, "copy_ceiling", "frac")), "kernel_us": _num(_dig(r, "copy_ceiling", "kernel_us"))},
                "position_only_frac": _num(_dig(r, "position_only", "frac"))}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data", "timed_steps", "repeats")}
    out["value"], out["ms_per_step"] = _num(out["value"], 7), _num(out["ms_per_step"], 6)
    out["config"] = {k: cfg.get(k) for k in ("workload", "sharding", "n_ranks", "process_group", "sets") if k in cfg}
    out["roofline"] = roof
    par = full.get("parity")
    out["parity"] = None if not isinstance(par, dict) else {k: _num(par.get(k)) for k in ("max_rel_err", "bit_exact", "checked_vertices")}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": (cb.get("sample") or "")[:160], "omp_value": _num(cb.get("omp_value")), "omp_cores": cb.get("omp_cores")}
    # N > 1: the other scalings beside `value`
    for k in ("weak_value", "weak_ms_per_step", "strong_value", "strong_ms_per_step", "value_with_gather", "value_with_gather_form",
              "strong_with_gather_value", "strong_with_gather_form", "crowd_value", "crowd_frame_ms", "comm_error"):
        if k in full:
            v = full[k]
            out[k] = v[:200] if isinstance(v, str) else _num(v, 7)
    legs = full.get("exchange_legs") or _dig(ex, "strong_scaling")
    if isinstance(legs, dict):
        dl_ = {}
        for k, v in legs.items():
            if isinstance(v, dict) and "value" in v and k != "compute_only":
                dl_[k] = {"value": _num(v.get("value"), 7), "ms_per_step": _num(v.get("ms_per_step"))}
        if dl_:
            out["exchange_legs"] = dl_
        if legs.get("gathered_equals_oracle") is not None:
            out["gathered_equals_oracle"] = legs.get("gathered_equals_oracle")
    dg = {}

    def put(key, *path):
        v = _dig(ex, *path)
        if v is not None:
            dg[key] = _num(v)
    for c in ("c2", "c5", "c3", "c3_root_motion"):
        put(f"{c}_frame_ms", c, "frame_ms")
        put(f"{c}_frame_ms_one_stream", c, "frame_ms_one_stream")
        put(f"{c}_skin_ms", c, "skin_ms")
        put(f"{c}_pose_ms", c, "pose_ms")
        put(f"{c}_bit_exact_frames", c, "parity", "frames_in_lock_step")
        put(f"{c}_max_rel_err", c, "parity", "end_to_end_max_rel_err")
    put("c3_crowd_kernel_us", "c3", "roofline", "kernel_us")
    put("c3_crowd_kernel_frac", "c3", "roofline", "frac")
    put("c3_fused_frac", "c3_fused", "roofline", "frac")
    put("c3_fused_max_rel_err", "c3_fused", "parity", "max_rel_err")
    put("c4_random_bones_frac", "c4_random_bones", "frac")
    put("c4_random_bones_slowdown", "c4_random_bones", "slowdown_vs_coherent")
    put("c3_random_bones_slowdown", "c3_random_bones", "slowdown_vs_coherent")
    for s in ("scene_256x1", "scene_64x4"):
        put(f"{s}_frame_ms", s, "frame_ms")
        put(f"{s}_frame_ms_pipelined", s, "frame_ms_pipelined")
        put(f"{s}_host_ms", s, "host_ms_skin_outputs")
        put(f"{s}_pose_ms", s, "pose_ms")
        put(f"{s}_skin_ms", s, "skin_ms")
        put(f"{s}_skin_frac", s, "skin_roofline", "frac")
        put(f"{s}_bit_exact", s, "parity", "bit_exact")
    for key, name in (("vb_plain", "plain"), ("vb_4_shapes", "with_4_blend_shapes")):
        put(f"{key}_frac", "vertex_buffer", name, "roofline", "frac")
        put(f"{key}_frac_two_streams", "vertex_buffer", name, "roofline", "frac_two_streams")
        put(f"{key}_frac_at_4_sets", "vertex_buffer", name, "roofline", "by_number_of_rotating_sets", "4", "frac")
    put("c3_plan_us_with_memo", "host_control_plane", "c3_plan_us_with_memo")
    put("scene_256x1_plan_us", "host_control_plane", "scene_256x1_plan_us")
    if dg:
        out["digest"] = dg
    b = full.get("box")
    if isinstance(b, dict):
        ah = b.get("after_headline") or b.get("at_end") or b.get("at_start") or {}
        out["box"] = {"sclk_under_load": _dig(ah, "sclk", "current"), "mclk": _dig(ah, "mclk", "current"),
                      "partition": f"{ah.get('current_compute_partition')}/{ah.get('current_memory_partition')}",
                      "power_w_under_load": None if ah.get("hwmon_power1_input") is None else ah["hwmon_power1_input"] // 1_000_000,
                      "power_cap_w": None if ah.get("hwmon_power1_cap") is None else ah["hwmon_power1_cap"] // 1_000_000,
                      "cus": _dig(b, "hip_device", "multi_processor_count")}
    out["full_record"] = full_path
    return out


def compact_line(full: dict, full_path: str | None) -> str:
    """json of compact_record, never longer than LINE_LIMIT: if a future key pushes it over, the digest goes first, then the box."""
    rec = compact_record(full, full_path)
    line = json.dumps(rec, separators=(",", ":"))
    for victim in ("digest", "box", "exchange_legs"):
        if len(line) <= LINE_LIMIT:
            break
        rec.pop(victim, None)
        rec["dropped_for_length"] = rec.get("dropped_for_length", []) + [victim]
        line = json.dumps(rec, separators=(",", ":"))
    return line


def finish(full: dict, full_path: str | None = None) -> None:
    """Rank 0's last act: the full record to `full_path` (default bench_full.json beside this file, and a copy under gpurun_out/ when that
    exists), then the compact line as the LAST thing on stdout."""
    path = full_path or os.path.join(ROOT, "bench_full.json")
    written = None
    for p in (path, os.path.join(ROOT, "gpurun_out", os.path.basename(path))):
        try:
            if os.path.isdir(os.path.dirname(p)):
                with open(p, "w") as f:
                    json.dump(full, f)
                written = written or os.path.relpath(p, ROOT)
        except Exception as e:     # noqa: BLE001
            print(f"# could not write {p}: {e!r}", file=sys.stderr)
    emit(compact_line(full, written))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--sets", type=int, default=8, help="disjoint buffer sets rotated through")
    ap.add_argument("--verts", type=int, default=N_VERTS)
    ap.add_argument("--bones", type=int, default=N_BONES)
    ap.add_argument("--random-bones", action="store_true", help="fully random bone indices (worst-case LDS gather)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="what `value` is at N > 1.  Default: strong = BASELINE config 4 as written, ONE 1 M-vertex mesh cut by vertex range over the N "
                         "GPUs, compute only (weak_value and value_with_gather beside it); weak = every GPU skins its own 1 M-vertex mesh.  N = 1: the same thing")
    ap.add_argument("--allgather", action="store_true", help="strong scaling: include the RCCL exchange in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--opt", action="append", default=[], help="kernel option key=value (e.g. lbs.dyn=0)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity spot-check before timing")
    ap.add_argument("--no-extras", action="store_true", help="skip the C2 / C3 / C5 sub-records")
    ap.add_argument("--extras-only", action="store_true", help="(internal) run only the sub-records, in this fresh process, and print them")
    ap.add_argument("--max-repeats", type=int, default=4000)
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic now (two rocprofv3 --pmc passes of ~15 s in child processes); replay the builder's")
    ap.add_argument("--one-process", action="store_true",
                    help="N > 1 without a launcher and without torch.distributed: ONE process drives all N GPUs, one fyx context and one "
                         "host thread per GPU, the exchange through fyx_comm_init_all / fyx_allgather_skinned_all.  Taken automatically when "
                         "the re-execution under torch.distributed.run fails or does not finish")
    ap.add_argument("--launcher-timeout", type=float, default=1500.0, help="seconds the plain `--gpus N` launch gives torch.distributed.run")
    ap.add_argument("--full-record", default=None, help="where the FULL record goes (default bench_full.json beside bench.py); the one stdout line is its digest")
    return ap.parse_args()


def cpu_baseline(mesh, pal, seconds: float) -> dict:
    """The oracle (C restatement of the reference's SERIAL loop, mesh/mod.rs:501-522 + the shader's
    normal/tangent math) timed on this host, 1 thread, on whole passes over the same workload."""
    import oracle
    oracle.lib()
    t0 = time.perf_counter()
    passes = 0
    while True:
        oracle.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal, mesh.normal, mesh.tangent, threads=1)
        passes += 1
        el = time.perf_counter() - t0
        if (el >= seconds and passes >= 2) or passes >= 1000:
            break
    serial = passes * mesh.n_verts / el
    # generous upper bound that does NOT exist in the reference: same arithmetic, OpenMP over vertices
    nthr = oracle.omp_max_threads()
    t0 = time.perf_counter()
    p2 = 0
    while True:
        oracle.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal, mesh.normal, mesh.tangent, threads=0)
        p2 += 1
        el2 = time.perf_counter() - t0
        if (el2 >= seconds / 2 and p2 >= 2) or p2 >= 5000:
            break
    return {"value": serial, "unit": "vertices/s", "cores": 1, "kind": "port",
            "sample": f"{passes} full passes over the same {mesh.n_verts}-vertex/{pal.shape[0]}-bone workload "
                      f"({el:.1f} s), C restatement of Fyrox's serial CPU loop (Rust toolchain unavailable)",
            "omp_value": p2 * mesh.n_verts / el2, "omp_cores": nthr,
            "omp_note": "OpenMP over vertices; not present in the reference (no rayon on this path)"}


def dl(ctx, t, first: int, count: int) -> np.ndarray:
    """`count` float32 of a torch device tensor from element `first`, through the library's own copy (fyx_memcpy_d2h on the
    context stream).  torch's .cpu() / .zero_() go through the legacy default stream; after the first of them every stream and
    event call of the process got slower (a host-bound frame loop lost 15 %), so the bench keeps torch to allocation."""
    out = np.empty(count, np.float32)
    ctx._check(ctx._l.fyx_memcpy_d2h(ctx._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(t.data_ptr() + 4 * first), out.nbytes))
    return out


def zero(ctx, t) -> None:
    z = np.zeros(t.numel(), np.float32)
    ctx._check(ctx._l.fyx_memcpy_h2d(ctx._h, ctypes.c_void_p(t.data_ptr()), z.ctypes.data_as(ctypes.c_void_p), z.nbytes))


def lbs_parity(ctx, mesh, pal, d_out, n_chk: int) -> dict:
    """Skinned position / normal / tangent of the first n_chk vertices against the oracle (checker only)."""
    import oracle
    ref = oracle.lbs_skin(mesh.pos[:n_chk], mesh.weights[:n_chk], mesh.indices[:n_chk], pal,
                          mesh.normal[:n_chk], mesh.tangent[:n_chk], threads=CHECK_THREADS)
    got = {"pos": dl(ctx, d_out[0], 0, n_chk * 3).reshape(-1, 3), "normal": dl(ctx, d_out[1], 0, n_chk * 3).reshape(-1, 3),
           "tangent": dl(ctx, d_out[2], 0, n_chk * 4).reshape(-1, 4)}
    err = max(float(np.abs(got[k] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-3)) for k in ref)
    return {"max_rel_err": err, "bit_exact": bool(all(np.array_equal(got[k], ref[k]) for k in ref)),
            "streams_checked": ["pos", "normal", "tangent"], "checked_vertices": n_chk}


# ---- the other BASELINE configs, end to end (N = 1) -------------------------------------------------------------------

def _chain_record(ctx, name, sc, mesh, n_instances, frames, desync, parity_instances, inst_offset=0, fused=False):
    """pose (AnimationPlayer / Machine) -> palette (written by the update kernel) -> instanced skinning, all resident.
    Timing: HIP events over `frames` frames on ONE stream (the frame is a dependent chain).  Parity: the whole chain
    against the oracle stepped in lock-step (tests/anim_cases.py builds both sides from one description)."""
    import oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import anim_cases as cases
    from fyrox_amd import anim as A

    nb = sc.rig.n_nodes
    p = cases.build_product(ctx, sc, n_instances)
    base = p.base_id
    bone_nodes = list(range(nb))
    A.create_bone_list(ctx, base + 50, base, bone_nodes)
    d_pal = ctx.malloc(n_instances * nb * 64)
    d_pal2 = ctx.malloc(n_instances * nb * 64)      # the pipelined frame alternates its palette buffers
    p.set_palette_output(base + 50, d_pal.ptr)
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    nv = mesh.n_verts * n_instances
    d_pos, d_nrm, d_tan = ctx.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64])
    oracles = {}
    # every instance at its own phase (a crowd), a function of its GLOBAL index (inst_offset: this rank's first instance when the
    # crowd is cut by instance range over several GPUs); the sampled ones get an oracle of their own
    if desync:
        for i in range(n_instances):
            for a in range(len(sc.animations)):
                p.set_time_position(a, ((inst_offset + i) * 0.37 + a * 0.11) % 1.0, instance=i)
    for i in parity_instances:
        o = cases.build_oracle(orc, sc)
        if desync:
            for a in range(len(sc.animations)):
                orc._alib().fo_animation_set_time_position(o.anims[a], ((inst_offset + i) * 0.37 + a * 0.11) % 1.0)
        oracles[i] = o
    update = p.update_machine if sc.machine is not None else p.update_animations

    def frame(skin=True):
        update(sc.dt)
        if skin:
            ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)

    # parity: a few frames in lock-step with the oracle, then the whole chain compared
    n_par = 5
    for f in range(n_par):
        for o in oracles.values():
            (o.update_machine if sc.machine is not None else o.update_animations)(sc.dt)
        frame()
    ctx.sync()
    pal = d_pal.download(np.float32, n_instances * nb * 16).reshape(n_instances, nb, 16)
    got = {"pos": d_pos.download(np.float32, nv * 3).reshape(n_instances, -1, 3),
           "normal": d_nrm.download(np.float32, nv * 3).reshape(n_instances, -1, 3),
           "tangent": d_tan.download(np.float32, nv * 4).reshape(n_instances, -1, 4)}
    chain_err, lbs_exact, chain_exact = 0.0, True, True
    refs_gpu_pal = {}
    for i, o in oracles.items():
        ref_pal = o.palette(bone_nodes)
        ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, ref_pal, mesh.normal, mesh.tangent, threads=CHECK_THREADS)
        ref_gpu_pal = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal[i], mesh.normal, mesh.tangent, threads=CHECK_THREADS)
        for k in ref:
            chain_err = max(chain_err, float(np.abs(got[k][i] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-3)))
            chain_exact &= bool(np.array_equal(got[k][i], ref[k]))
            lbs_exact &= bool(np.array_equal(got[k][i], ref_gpu_pal[k]))     # the skinning stage given the GPU's palettes
        refs_gpu_pal[i] = ref_gpu_pal
        o.close()
    if chain_err > 1e-5:
        raise SystemExit(f"{name}: end-to-end parity failed: max rel err {chain_err:.3e} > 1e-5")
    if not lbs_exact:
        raise SystemExit(f"{name}: skinning stage is not bit-exact against the oracle on the GPU-built palettes")

    # the first few dozen frames after the (CPU-heavy) parity leg run slower than the steady state: long warm-up
    for _ in range(60):
        frame()
    ctx.sync()
    ctx.timer_begin()
    for _ in range(frames):
        frame()
    frame_serial_ms = ctx.timer_end() / frames
    # pipelined: whole frames alternate between two streams (option anim.overlap), so frame n + 1's pose kernels run beside frame n's
    # skinning; a PAIR of palette buffers registered once (the frames of the two streams write one each) and two sets of vertex
    # outputs (consecutive frames' skinning launches are not ordered against each other: a renderer draws frame n from one set
    # while frame n + 1 is skinned into the other) -- nothing else changes, the same kernels compute the same values
    ctx.set_option("anim.overlap", 1)
    pals = (d_pal, d_pal2)
    outs2 = tuple(ctx.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64]))
    out_sets = ((d_pos, d_nrm, d_tan), outs2)
    p.set_palette_output_pair(base + 50, d_pal.ptr, d_pal2.ptr)

    def frame_pipelined(k):     # frame k of the mode runs on stream (k + 1) & 1 (the mode's first frame starts on the second stream)
        dp, o = pals[(k + 1) & 1], out_sets[(k + 1) & 1]
        update(sc.dt)
        ctx.lbs_skin_device(base + 60, dp.ptr, nb, n_instances, o[0].ptr, o[1].ptr, o[2].ptr)

    for k in range(20):
        frame_pipelined(k)
        if p.current_palette(base + 50) != pals[(k + 1) & 1].ptr:
            raise SystemExit(f"{name}: pipelined frame {k} wrote the other palette buffer of the pair")
    ctx.sync()
    ctx.timer_begin()
    for k in range(frames):
        frame_pipelined(k)
    frame_ms = ctx.timer_end() / frames
    ctx.set_option("anim.overlap", 0)
    p.set_palette_output(base + 50, d_pal.ptr)
    ctx.sync()
    # the pipelined frames computed what the serial ones compute: the last frame's vertices against fyx_lbs_skin_device on the palette it wrote
    last = (20 + frames - 1 + 1) & 1
    got_p = [b.download(np.uint32, nv * w) for b, w in zip(out_sets[last], (3, 3, 4))]
    ctx.lbs_skin_device(base + 60, pals[last].ptr, nb, n_instances, *(b.ptr for b in out_sets[last ^ 1]))
    ctx.sync()
    if not all(np.array_equal(x, b.download(np.uint32, nv * w)) for x, b, w in zip(got_p, out_sets[last ^ 1], (3, 3, 4))):
        raise SystemExit(f"{name}: the pipelined frames' vertices differ from a skinning launch on the palette they wrote")
    del got_p
    for b in outs2:
        b.free()
    frame_by_kind_ms = None      # (the streams-by-kind mode is a debug option since round 6: it measured the one-stream time for crowds)
    # the same with the mesh registered as the animator's skin output: the update call is the frame, ONE set of vertex outputs, and the
    # library orders frame n + 1's skinning launch behind frame n's (its pose kernels still run beside frame n's skinning)
    frame_pipelined_registered_ms, registered_identical = None, None
    if n_instances >= 4:
        ctx.set_option("anim.overlap", 1)
        p.set_palette_output_pair(base + 50, d_pal.ptr, d_pal2.ptr)
        p.set_skin_output(base + 50, base + 60, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
        for _ in range(20):
            update(sc.dt)
        ctx.sync()
        ctx.timer_begin()
        for _ in range(frames):
            update(sc.dt)
        frame_pipelined_registered_ms = ctx.timer_end() / frames
        cur = p.current_palette(base + 50)
        a = [d_pos.download(np.uint32, nv * 3), d_nrm.download(np.uint32, nv * 3), d_tan.download(np.uint32, nv * 4)]
        ctx.set_option("anim.overlap", 0)
        p.set_skin_output(base + 50, base + 60)
        p.set_palette_output(base + 50, d_pal.ptr)
        ctx.lbs_skin_device(base + 60, cur, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
        ctx.sync()
        b = [d_pos.download(np.uint32, nv * 3), d_nrm.download(np.uint32, nv * 3), d_tan.download(np.uint32, nv * 4)]
        registered_identical = all(bool(np.array_equal(x, y)) for x, y in zip(a, b))
        if not registered_identical:
            raise SystemExit(f"{name}: pipelined frames with a registered skin output differ from fyx_lbs_skin_device on the palette they wrote")
        del a, b
    # one character: the frame as ONE launch through to the vertices (fyx_animator_set_skin_output: the pose launch also holds the
    # skinning workgroups, which form the palette on chip) -- the update call is the whole frame
    frame_one_launch_ms, one_launch_identical = None, None
    if n_instances < 4:
        p.set_skin_output(base + 50, base + 60, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
        for _ in range(60):
            update(sc.dt)
        ctx.sync()
        ctx.timer_begin()
        for _ in range(frames):
            update(sc.dt)
        frame_one_launch_ms = ctx.timer_end() / frames
        a = [d_pos.download(np.uint32, nv * 3), d_nrm.download(np.uint32, nv * 3), d_tan.download(np.uint32, nv * 4)]
        p.set_skin_output(base + 50, base + 60)
        ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
        ctx.sync()
        b = [d_pos.download(np.uint32, nv * 3), d_nrm.download(np.uint32, nv * 3), d_tan.download(np.uint32, nv * 4)]
        one_launch_identical = all(bool(np.array_equal(x, y)) for x, y in zip(a, b))
        if not one_launch_identical:
            raise SystemExit(f"{name}: the one-launch frame's vertices differ from fyx_lbs_skin_device on the same palette")
    ctx.timer_begin()
    for _ in range(frames):
        frame(skin=False)
    pose_ms = ctx.timer_end() / frames
    ctx.timer_begin()
    for _ in range(frames):
        ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
    skin_ms = ctx.timer_end() / frames
    ctx.set_option("lbs.timing", 1)      # the kernel's own duration: every launch with its own start / stop events
    ctx.kernel_time()
    for _ in range(frames):
        ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
    k_us, k_n = ctx.kernel_time()
    ctx.set_option("lbs.timing", 0)
    skin_kernel_us = k_us / max(k_n, 1)
    # the same dispatch INSIDE the frame (pose kernels before it on the stream): a launch that follows another skinning launch
    # directly inherits that launch's un-drained writes -- the kernel "ends" when the caches have accepted its 400 MB, not when
    # HBM has them -- and runs 10 - 20 % longer than one that follows ~50 us of pose kernels (profiles/r03_summary.json,
    # crowd_duration_spread_in_the_c3_frame_trace)
    ctx.set_option("lbs.timing", 1)
    ctx.kernel_time()
    for _ in range(frames):
        frame()
    kf_us, kf_n = ctx.kernel_time()
    ctx.set_option("lbs.timing", 0)
    skin_kernel_in_frame_us = kf_us / max(kf_n, 1)
    unique = mesh.n_verts * 60 + n_instances * nb * 64 + nv * 40     # mesh read once, palettes, outputs
    fused_rec = None
    if fused:     # the same launch with lbs.exact = 0 (FMA; the crowd kernel blends the four matrices first): inside north_star's 1e-5
        ctx.set_option("lbs.exact", 0)
        ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
        ctx.sync()
        gf = {"pos": d_pos.download(np.float32, nv * 3).reshape(n_instances, -1, 3), "normal": d_nrm.download(np.float32, nv * 3).reshape(n_instances, -1, 3),
              "tangent": d_tan.download(np.float32, nv * 4).reshape(n_instances, -1, 4)}
        pal_now = d_pal.download(np.float32, n_instances * nb * 16).reshape(n_instances, nb, 16)     # the palettes have moved on since the lock-step frames
        f_err = 0.0
        for i in refs_gpu_pal:
            r = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal_now[i], mesh.normal, mesh.tangent, threads=CHECK_THREADS)
            f_err = max(f_err, max(float(np.abs(gf[k][i] - r[k]).max() / max(np.abs(r[k]).max(), 1e-3)) for k in r))
        if f_err > 1e-5:
            raise SystemExit(f"{name}: fused mode outside 1e-5: max rel err {f_err:.3e}")
        for _ in range(30):
            ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
        ctx.sync()
        ctx.timer_begin()
        for _ in range(frames):
            ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
        f_ms = ctx.timer_end() / frames
        ctx.set_option("lbs.timing", 1)
        ctx.kernel_time()
        for _ in range(frames):
            ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
        fk_us, fk_n = ctx.kernel_time()
        ctx.set_option("lbs.timing", 0)
        ctx.set_option("lbs.exact", 1)
        fk = fk_us / max(fk_n, 1)
        fused_rec = {"workload": name + " -- the skinning launch alone, lbs.exact = 0 (fused multiply-adds, matrices blended first)",
                     "skin_ms": f_ms, "skinned_vertices_per_s_skin": nv / (f_ms * 1e-3),
                     "roofline": {"bound": "hbm", "kernel": "lbs_skin_crowd" if n_instances >= 4 else "lbs_skin", "unique_bytes_per_launch": unique,
                                  "achieved": unique / (fk * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": unique / (fk * 1e-6) / 1e9 / HBM_PEAK_GBPS, "kernel_us": fk, "launch_period_us": f_ms * 1e3},
                     "parity": {"instances_checked": sorted(refs_gpu_pal), "max_rel_err": f_err, "tolerance": 1e-5, "bit_exact": False,
                                "note": "against the oracle on the GPU-built palettes; north_star allows 1e-5 relative"}}
    modes = {"pipelined": frame_ms, "one_stream": frame_serial_ms}
    if frame_pipelined_registered_ms is not None:
        modes["pipelined_registered_skin_output"] = frame_pipelined_registered_ms
    if frame_by_kind_ms is not None:
        modes["pipelined_streams_by_kind"] = frame_by_kind_ms
    if frame_one_launch_ms is not None:
        modes["one_launch"] = frame_one_launch_ms
    best_mode = min(modes, key=modes.get)
    best_ms = modes[best_mode]
    rec = {"workload": name, "frame_ms": best_ms, "frame_mode": best_mode,
           "frame_ms_pipelined": frame_ms, "frame_ms_one_stream": frame_serial_ms, "frame_ms_one_launch": frame_one_launch_ms,
           "frame_ms_pipelined_registered_skin_output": frame_pipelined_registered_ms, "frame_ms_pipelined_streams_by_kind": frame_by_kind_ms,
           "pipelined_registered_vertices_bit_identical_to_lbs_skin": registered_identical,
           "one_launch_vertices_bit_identical_to_lbs_skin": one_launch_identical, "pose_ms": pose_ms, "skin_ms": skin_ms,
           "frame_over_skin": best_ms / skin_ms,
           "frame_roofline_frac": unique / (best_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "frame_note": "pipelined: whole frames alternate between two streams (anim.overlap = 1, a palette pair registered once, TWO sets of vertex "
                         "outputs: the caller's skinning launches of consecutive frames are not ordered against each other): frame n + 1's pose kernels "
                         "run beside frame n's skinning; pipelined_registered_skin_output: the same with the mesh registered as the animator's skin "
                         "output -- the update call is the frame, ONE set of vertex outputs, the library orders frame n + 1's skinning behind frame n's; "
                         "pipelined_streams_by_kind: anim.overlap = 2 -- pose kernels on one stream, skinning launches on another, each in order, ONE set "
                         "of vertex outputs; "
                         "one_stream: the whole frame as one dependent chain on one stream; one_launch (one character): "
                         "fyx_animator_set_skin_output -- sampler, update and skinning workgroups in ONE launch, the update call is the frame; frame_ms is "
                         "the fastest (the host picks the mode per scene); frame_roofline_frac = the skinning's unique bytes / frame_ms / 8 TB/s: "
                         "the frame end to end against the HBM roofline",
           "skinned_vertices_per_s_frame": nv / (best_ms * 1e-3), "skinned_vertices_per_s_skin": nv / (skin_ms * 1e-3),
           "roofline": {"bound": "hbm", "kernel": "lbs_skin_crowd" if n_instances >= 4 else "lbs_skin",
                        "unique_bytes_per_launch": unique, "achieved": unique / (skin_kernel_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": unique / (skin_kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                        "kernel_us": skin_kernel_us,
                        "kernel_us_note": "the skinning dispatch's own duration (per-dispatch events) when nothing but skinning launches run, back to back "
                                          "on one stream: the definition of rounds 1 - 2 and the conservative one (each launch inherits its predecessor's "
                                          "write-back: a kernel ends when the caches have accepted its stores, not when HBM has them)",
                        "kernel_us_in_frame": skin_kernel_in_frame_us,
                        "frac_in_frame": unique / (skin_kernel_in_frame_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                        "in_frame_note": "the same dispatch INSIDE the one-stream frame loop of this workload, i.e. behind the frame's pose kernels -- what a "
                                         "kernel trace of the frame reports for it (round 3 quoted this one as `frac`)",
                        "launch_period_us": skin_ms * 1e3},
           "parity": {"instances_checked": sorted(oracles), "frames_in_lock_step": n_par,
                      "end_to_end_max_rel_err": chain_err, "end_to_end_bit_exact": chain_exact,
                      "bit_exact": lbs_exact,
                      "note": "bit_exact = skinning stage vs oracle on the GPU-built palettes; end_to_end = pose -> palette -> "
                              "skin vs the oracle's whole chain (Euler tracks use device sincosf: <= 1e-5, not bit-exact)"}}
    if fused_rec is not None:
        rec["fused"] = fused_rec
    for b in (d_pal, d_pal2, d_pos, d_nrm, d_tan):
        b.free()
    ctx.mesh_free(base + 60)
    p.free()
    return rec


def _c3_record(ctx, n_instances=1000, inst_offset=0, frames=300, parity_instances=(0, 1, 15, 16, 17, 999), fused=True, root_motion=False):
    """BASELINE config 3 (the crowd), or this rank's instance range of it.  root_motion: RootMotionSettings on every clip (root = node
    0, nothing ignored) and AnimationPose::root_motion tracked through the machine (lib.rs:498-661): two more kernels per frame and the
    planner's root-motion programs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import anim_cases as cases
    from fyrox_amd import synth
    par = sorted({i for i in parity_instances if 0 <= i < n_instances})
    sc = cases.c5_blend_tree(n_bones=64, seed=synth.SEED_BASE + 3)
    name = f"C3: crowd of {n_instances} instances x 10k verts / 64 bones, 4-clip blend-tree machine per instance"
    if root_motion:
        for a in sc.animations:
            a.root_motion = (0, False, False, False, False)
        sc.track_root_motion = True
        name += ", root motion on every clip"
    return _chain_record(ctx, name, sc, synth.make_mesh(10_000, 64, synth.SEED_BASE + 3),
                         n_instances, frames, True, par, inst_offset=inst_offset, fused=fused)


def _c3_random_record(ctx, n_instances=1000, n_verts=10_000, n_bones=64, launches=200) -> dict:
    """C3's skinning launch with FULLY RANDOM bone indices (SURVEY 8(d) worst case) against the coherent mesh, same palettes, same
    buffers, same process: the crowd kernel's own duration (per-dispatch events), instances 0 and 999 against the oracle."""
    import oracle as orc
    from fyrox_amd import synth
    seed = synth.SEED_BASE + 3
    pal = synth.make_palette(n_bones, seed, n_instances=n_instances).reshape(n_instances, n_bones, 16)
    d_pal = ctx.to_device(pal)
    nv = n_verts * n_instances
    outs = tuple(ctx.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64]))
    unique = n_verts * 60 + n_instances * n_bones * 64 + nv * 40
    res = {}
    for key, coherent in (("coherent", True), ("random", False)):
        mesh = synth.make_mesh(n_verts, n_bones, seed, coherent=coherent)
        ctx.mesh_upload_soa(950, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)

        def launch():
            ctx.lbs_skin_device(950, d_pal.ptr, n_bones, n_instances, outs[0].ptr, outs[1].ptr, outs[2].ptr)

        launch()
        ctx.sync()
        exact = True
        for i in (0, n_instances - 1):
            ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal[i], mesh.normal, mesh.tangent, threads=CHECK_THREADS)
            got = dl(ctx, _Ptr(outs[0].ptr), i * n_verts * 3, n_verts * 3).reshape(-1, 3)
            exact &= bool(np.array_equal(got, ref["pos"]))
        if not exact:
            raise SystemExit(f"C3, {key} bone indices: the crowd launch differs from the oracle")
        for _ in range(20):
            launch()
        ctx.set_option("lbs.timing", 1)
        ctx.kernel_time()
        for _ in range(launches):
            launch()
        us, n = ctx.kernel_time()
        ctx.set_option("lbs.timing", 0)
        k = us / max(n, 1)
        res[key] = {"kernel_us": k, "frac": unique / (k * 1e-6) / 1e9 / HBM_PEAK_GBPS, "bit_exact": exact}
        ctx.mesh_free(950)
    for b in outs:
        b.free()
    d_pal.free()
    return {"workload": f"C3's skinning launch ({n_instances} x {n_verts} verts / {n_bones} bones, synthetic palettes) with coherent and with FULLY RANDOM bone "
                        "indices (SURVEY 8(d) worst case), the kernel's own duration back to back",
            **res, "slowdown_vs_coherent": res["random"]["kernel_us"] / res["coherent"]["kernel_us"] - 1.0}


class _Ptr:
    """Stands in for a torch tensor where dl() only needs the address."""
    def __init__(self, p):
        self._p = p

    def data_ptr(self):
        return self._p


class _Out:
    """An output stream of its own allocation (fyx_malloc_streams), with the two methods of a torch tensor dl() / zero() use."""
    def __init__(self, buf, n_floats):
        self.buf, self._n = buf, n_floats

    def data_ptr(self):
        return self.buf.ptr

    def numel(self):
        return self._n


def output_streams(ctx, n_verts: int) -> tuple:
    """Position / normal / tangent outputs of n_verts vertices, each stream an allocation of its own (include/fyrox_hip.h, fyx_malloc_streams)."""
    p, n, t = ctx.malloc_streams([n_verts * 12 + 64, n_verts * 12 + 64, n_verts * 16 + 64])
    return (_Out(p, n_verts * 3 + 16), _Out(n, n_verts * 3 + 16), _Out(t, n_verts * 4 + 16))


def _vertex_buffer_record(ctx, n_verts=1_000_000, n_bones=256, n_shapes=4, sets=6, steps=400) -> dict:
    """The engine's own vertex format in, the same format out: the mesh is uploaded as the interleaved 68-byte
    AnimatedVertex stream (scene/mesh/vertex.rs:139-155, buffer.rs:404-415) and fyx_lbs_skin_ex writes a render-ready vertex buffer of
    the same layout (position / normal / tangent.xyz replaced, everything else passed through) -- lbs_skin_aos, whole 68-byte records
    in and out: 2 x stride = 136 B per vertex, + 18 B per vertex and blend shape (three f16 offsets x 3).  Launch period (ex launches
    carry no per-dispatch events), one and two launch streams; byte-level parity of the first 20 000 vertices against the oracle."""
    import oracle as orc
    from fyrox_amd import synth
    L = synth.ANIMATED_VERTEX
    seed = synth.SEED_BASE + 4
    mesh = synth.make_mesh(n_verts, n_bones, seed)
    palh = synth.make_palette(n_bones, seed)
    pal = ctx.to_device(palh)
    storage, plane, w = synth.make_blend_shapes(n_verts, n_shapes, seed)
    d_w = ctx.to_device(w)
    aos = mesh.to_animated_vertex_aos()
    outs = []
    for k in range(sets):
        ctx.mesh_upload(900 + k, aos, n_verts, L["stride"], off_pos=L["off_pos"], off_normal=L["off_normal"], off_tangent=L["off_tangent"],
                        off_weights=L["off_weights"], off_indices=L["off_indices"])
        ctx.mesh_set_blend_shapes(900 + k, storage, n_shapes, plane)
        outs.append(ctx.malloc(n_verts * L["stride"] + 256))

    def launch(k, shapes):
        ctx.lbs_skin_ex(900 + k, pal.ptr, n_bones, 1, d_blend_shape_weights=d_w.ptr if shapes else 0, n_blend_shapes=n_shapes if shapes else 0,
                        d_out_vertices=outs[k].ptr, out_stride=0)

    n_chk = 20_000
    src = aos.reshape(n_verts, L["stride"])[:n_chk]
    res = {}
    for shapes in (False, True):
        launch(0, shapes)
        ctx.sync()
        raw = outs[0].download(np.uint8, n_chk * L["stride"]).reshape(n_chk, L["stride"])
        p_, n_, t_ = mesh.pos[:n_chk], mesh.normal[:n_chk], mesh.tangent[:n_chk]
        if shapes:
            p_, n_, t_ = orc.apply_blend_shapes(mesh.pos, mesh.normal, mesh.tangent, storage, plane, w)
            p_, n_, t_ = p_[:n_chk], n_[:n_chk], t_[:n_chk]
        ref = orc.lbs_skin(p_, mesh.weights[:n_chk], mesh.indices[:n_chk], palh, n_, t_, threads=CHECK_THREADS)
        exact, touched = True, np.zeros(L["stride"], bool)
        for off, key in ((L["off_pos"], "pos"), (L["off_normal"], "normal"), (L["off_tangent"], "tangent")):
            got = np.ascontiguousarray(raw[:, off:off + 12]).view(np.uint32)
            exact &= bool(np.array_equal(got, np.ascontiguousarray(ref[key][:, :3]).view(np.uint32)))
            touched[off:off + 12] = True
        exact &= bool(np.array_equal(raw[:, ~touched], src[:, ~touched]))       # uv, tangent.w, weights, indices pass through
        if not exact:
            raise SystemExit("vertex-buffer record: the output vertex buffer differs from the oracle")
        bpv = 2 * L["stride"] + (18 * n_shapes if shapes else 0)
        per = {}
        for streams in (1, 2):
            ctx.set_option("lbs.streams", streams)
            for i in range(30):
                launch(i % sets, shapes)
            ctx.sync()
            ctx.timer_begin()
            for i in range(steps):
                launch(i % sets, shapes)
            per[streams] = ctx.timer_end() * 1e3 / steps
        ctx.set_option("lbs.streams", 1)
        us = per[1]
        # the same lone launches against the number of rotating sets (the footprint the rotation walks over: see roofline.by_number_of_rotating_sets
        # of the headline -- six sets of this record are 0.8 - 1.25 GB, past the knee)
        by_sets = {}
        for k in (1, 2, 3, 4, sets):
            for i in range(20):
                launch(i % k, shapes)
            ctx.sync()
            ctx.timer_begin()
            for i in range(steps):
                launch(i % k, shapes)
            t_us = ctx.timer_end() * 1e3 / steps
            by_sets[str(k)] = {"launch_period_us": t_us, "frac": bpv * n_verts / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "footprint_MB": k * bpv * n_verts / 1e6}
        res["with_%d_blend_shapes" % n_shapes if shapes else "plain"] = {
            "algorithmic_bytes_per_vertex": bpv, "launch_period_us_one_stream": us, "launch_period_us_two_streams": per[2],
            "vertices_per_s": n_verts / (per[2] * 1e-6),
            "roofline": {"bound": "hbm", "kernel": "lbs_skin_aos", "algorithmic_bytes_per_launch": bpv * n_verts,
                         "achieved": bpv * n_verts / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": bpv * n_verts / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                         "frac_two_streams": bpv * n_verts / (per[2] * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                         "by_number_of_rotating_sets": by_sets,
                         "note": "launch period on one stream (includes the dependent-launch gap) over all the sets; frac_two_streams: launches overlapped; "
                                 "by_number_of_rotating_sets: the lone launches over the first k sets (footprint k x bytes per launch)"},
            "parity": {"checked_vertices": n_chk, "bit_exact": True, "note": "every byte of the first vertices of the output vertex buffer: skinned "
                       "attributes against the oracle, all other bytes against the input"}}
    for k in range(sets):
        ctx.mesh_free(900 + k)
        outs[k].free()
    pal.free(); d_w.free()
    return {"workload": f"vertex buffer in / vertex buffer out: {n_verts} verts / {n_bones} bones, 68-byte AnimatedVertex (vertex.rs:139-155), "
                        f"{sets} rotating sets", **res}


def _scene_record(ctx, n_chars: int, n_inst: int, n_verts: int, id_base: int, frames: int = 150) -> dict:
    """The scene tick: n_chars DISTINCT characters (own rig, four clips, blend-tree machine, mesh), n_inst instances
    each; per frame ONE fyx_scene_update (one launch per stage for all animators, palettes written by the update kernel) and ONE
    fyx_lbs_skin_batch.  Parity: two characters against the oracle after a few frames (skinning stage bit-exact)."""
    import ctypes as ct
    import oracle as orc
    from fyrox_amd import anim as A, synth
    from fyrox_amd._native import SkinJob
    nb, dt = 64, 1.0 / 60.0
    chars, frees = [], []
    for k in range(n_chars):
        seed = synth.SEED_BASE + 700 + k
        rig = synth.make_rig(nb, seed)
        rid, aid, bid, mid, tid = (id_base + j * 10_000 + k for j in range(5))
        tid = id_base + 100_000 + 4 * k
        A.create_rig(ctx, rid, rig)
        an = A.Animator(ctx, aid, rid, rig, n_inst)
        for c in range(4):
            td, tgt = synth.make_clip(nb, seed, clip=c, euler_every=10 ** 9)
            A.upload_tracks_data(ctx, tid + c, td)
            an.add_animation(tid + c, tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
        an.set_machine(synth.make_c5_machine())
        A.create_bone_list(ctx, bid, rid, list(range(nb)))
        mesh = synth.make_mesh(n_verts, nb, seed)
        ctx.mesh_upload_soa(mid, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        nv = n_verts * n_inst
        d_pal = ctx.malloc(n_inst * nb * 64)
        outs = tuple(ctx.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64]))
        an.set_palette_output(bid, d_pal.ptr)
        an.bones_id = bid
        chars.append((an, mid, d_pal, outs, mesh, rig, seed))
        frees += [d_pal, *outs]
    ids = np.asarray([c[0].id for c in chars], np.uint64)
    ids_p, cdt = ids.ctypes.data_as(ct.c_void_p), ct.c_float(dt)
    jobs = (SkinJob * n_chars)(*[SkinJob(mid, d_pal.ptr, nb, n_inst, o[0].ptr, o[1].ptr, o[2].ptr) for _, mid, d_pal, o, *_ in chars])
    upd, batch = ctx._l.fyx_scene_update, ctx._l.fyx_lbs_skin_batch

    def frame(pose=True, skin=True):
        if pose:
            ctx._check(upd(ctx._h, ids_p, n_chars, cdt))
        if skin:
            ctx._check(batch(ctx._h, jobs, n_chars))

    n_par = 4
    for _ in range(n_par):
        frame()
    ctx.sync()
    exact = True
    for k in (0, n_chars - 1):       # the oracle steps the same machine (all instances start in phase); skinning on the GPU's palette
        an, mid, d_pal, outs, mesh, rig, seed = chars[k]
        pal = d_pal.download(np.float32, n_inst * nb * 16).reshape(n_inst, nb, 16)
        o = orc.AnimScene(rig)
        for c in range(4):
            td, tgt = synth.make_clip(nb, seed, clip=c, euler_every=10 ** 9)
            o.add_animation(o.add_tracks_data(td), tgt, time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][c])
        o.set_machine(synth.make_c5_machine())
        for _ in range(n_par):
            o.update_machine(dt)
        ref_pal = o.palette(list(range(nb)))
        exact &= bool(np.array_equal(pal[0].view(np.uint32), ref_pal.view(np.uint32)))
        ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, ref_pal, mesh.normal, mesh.tangent, threads=CHECK_THREADS)
        got = outs[0].download(np.float32, n_verts * 3).reshape(-1, 3)
        exact &= bool(np.array_equal(got, ref["pos"]))
        o.close()
    if not exact:
        raise SystemExit("scene record: palettes / skinned vertices differ from the oracle")
    for _ in range(60):
        frame()

    def timed(**kw):
        ctx.sync()
        ctx.timer_begin()
        for _ in range(frames):
            frame(**kw)
        return ctx.timer_end() / frames

    f_ms, p_ms, s_ms = timed(), timed(skin=False), timed(pose=False)
    # what the calls cost the calling thread (issued without waiting; the frames queue up on the GPU): the frame is bound by the GPU side
    # when this is the smaller number
    def host_cost(**kw):      # a few frames issued into an EMPTY queue (a long run of them blocks on the queue's depth and measures the GPU)
        best = 1e9
        for _ in range(12):
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(4):
                frame(**kw)
            best = min(best, (time.perf_counter() - t0) * 1e3 / 4)
        ctx.sync()
        return best
    host_ms = host_cost()

    def host_sections_of(**kw):      # option debug.host_times: what fyx_scene_update's sections cost the calling thread, per frame
        ctx.set_option("debug.host_times", 1)
        ctx.host_times()
        for _ in range(200):
            frame(**kw)
        ctx.sync()
        ht = ctx.host_times()
        ctx.set_option("debug.host_times", 0)
        return {k: ht[i] / max(ht[6], 1.0) for i, k in enumerate(("plan_us", "state_and_jobs_us", "control_block_us", "stage_launches_us",
                                                                  "event_records_us", "skin_batch_us"))}
    # the same scene with the meshes registered as the animators' skin outputs (fyx_animator_set_skin_output): fyx_scene_update's update
    # launch also holds the skinning workgroups (they recompute their character's pose on chip) -- one call, one launch less, no
    # palette round trip; same bits (checked below against the batch's outputs)
    f2_ms, host2_ms, same, host_sections2 = None, None, None, None
    try:
        ref_out = [c_[3][0].download(np.uint32, n_verts * n_inst * 3) for c_ in (chars[0], chars[-1])]
        for an, mid, d_pal, outs, *_ in chars:
            an.set_skin_output(an.bones_id, mid, outs[0].ptr, outs[1].ptr, outs[2].ptr)
        for _ in range(30):
            frame(skin=False)
        f2_ms = timed(skin=False)
        host2_ms = host_cost(skin=False)
        host_sections2 = host_sections_of(skin=False)
        # parity of the mode: skin with the batch call on the palettes the last update wrote, compare with what that update skinned itself
        got = [c_[3][0].download(np.uint32, n_verts * n_inst * 3) for c_ in (chars[0], chars[-1])]
        for an, mid, d_pal, outs, *_ in chars:
            an.set_skin_output(an.bones_id, mid)
        frame(pose=False)
        ctx.sync()
        again = [c_[3][0].download(np.uint32, n_verts * n_inst * 3) for c_ in (chars[0], chars[-1])]
        same = all(bool(np.array_equal(x, y)) for x, y in zip(got, again))
        if not same:
            raise SystemExit("scene record: the update launch's own skinning differs from fyx_lbs_skin_batch on the same palettes")
        del ref_out
    except SystemExit:
        raise
    except Exception as e:     # noqa: BLE001
        f2_ms, same = None, repr(e)
    # pipelined scene frames (anim.overlap): whole frames alternate between two streams, every animator's palette output is a PAIR
    # registered once (fyx_animator_set_palette_output_pair: the frames of the two streams write one buffer each), the skin outputs
    # read the frame's own and the library orders frame n + 1's skinning of the vertex buffers behind frame n's.  Frame n + 1's
    # sampler / update launches run beside frame n's skinning launch.
    # anim.overlap = 2: streams by kind -- every pose kernel on the context stream, the skinning launches on the second stream, each in order.
    f3_ms, host3_ms, same3, host_sections = None, None, None, None
    f4_ms, host4_ms, host_sections4 = None, None, None
    try:
        pals2 = [ctx.malloc(n_inst * nb * 64) for _ in chars]
        frees += pals2
        for mode in (1,):      # (debug.overlap = 2, streams by kind, measured the same for scenes in round 5: no longer in the record)
            for (an, mid, d_pal, outs, *_), p2 in zip(chars, pals2):
                an.set_palette_output_pair(an.bones_id, d_pal.ptr, p2.ptr)
                an.set_skin_output(an.bones_id, mid, outs[0].ptr, outs[1].ptr, outs[2].ptr)
            ctx.set_option("anim.overlap", mode)
            for _ in range(30):
                frame(skin=False)
            t_ms, h_ms, h_sec = timed(skin=False), host_cost(skin=False), host_sections_of(skin=False)
            if mode == 1:
                f3_ms, host3_ms, host_sections = t_ms, h_ms, h_sec
            else:
                f4_ms, host4_ms, host_sections4 = t_ms, h_ms, h_sec
            # parity of the mode: the vertices the last pipelined frame skinned against the batch call on the palettes that frame wrote
            got = [c_[3][0].download(np.uint32, n_verts * n_inst * 3) for c_ in (chars[0], chars[-1])]
            cur = [an.current_palette(an.bones_id) for an, *_ in chars]
            ctx.set_option("anim.overlap", 0)
            for (an, mid, d_pal, *_) in chars:
                an.set_skin_output(an.bones_id, mid)
                an.set_palette_output(an.bones_id, d_pal.ptr)
            jobs_cur = (SkinJob * n_chars)(*[SkinJob(mid, pc, nb, n_inst, o[0].ptr, o[1].ptr, o[2].ptr) for (_, mid, _p, o, *_), pc in zip(chars, cur)])
            ctx._check(batch(ctx._h, jobs_cur, n_chars))
            ctx.sync()
            again = [c_[3][0].download(np.uint32, n_verts * n_inst * 3) for c_ in (chars[0], chars[-1])]
            same3 = all(bool(np.array_equal(x, y)) for x, y in zip(got, again))
            if not same3:
                raise SystemExit(f"scene record: the pipelined frames' vertices (anim.overlap = {mode}) differ from fyx_lbs_skin_batch on the palettes they wrote")
    except SystemExit:
        raise
    except Exception as e:     # noqa: BLE001
        same3 = repr(e)
        ctx.set_option("anim.overlap", 0)
    total = n_chars * n_inst * n_verts
    rec = {"workload": f"scene tick: {n_chars} distinct characters x {n_inst} instance(s) x {n_verts} verts / {nb} bones, 4-clip blend-tree machine each; "
                       "one fyx_scene_update + one fyx_lbs_skin_batch per frame",
           "frame_ms": min(f_ms, f2_ms) if f2_ms else f_ms, "frame_mode": "skin_outputs" if f2_ms and f2_ms < f_ms else "scene_update_then_skin_batch",
           "frame_ms_scene_update_then_skin_batch": f_ms, "frame_ms_skin_outputs": f2_ms, "skin_outputs_bit_identical_to_skin_batch": same,
           "frame_ms_pipelined": f3_ms, "pipelined_bit_identical_to_skin_batch": same3, "host_ms_pipelined": host3_ms,
           "frame_ms_pipelined_streams_by_kind": f4_ms, "host_ms_pipelined_streams_by_kind": host4_ms, "host_sections_pipelined_streams_by_kind_us": host_sections4,
           "pipelined_note": "anim.overlap = 1 with palette pairs and registered skin outputs: frames alternate between two streams, frame n + 1's "
                             "sampler / update launches run beside frame n's skinning launch; streams_by_kind: anim.overlap = 2, pose kernels on one "
                             "stream and skinning launches on another, each in order; frame_ms stays the one-stream frame",
           "host_sections_pipelined_us": host_sections, "host_sections_skin_outputs_us": host_sections2,
           "host_sections_note": "option debug.host_times: what the sections of fyx_scene_update cost the calling thread per frame while frames queue up "
                                 "(a section that has to wait for a control-block slot of a frame still in flight includes that wait)",
           "host_ms_scene_update_then_skin_batch": host_ms, "host_ms_skin_outputs": host2_ms,
           "gpu_side_ms": (min(f_ms, f2_ms) if f2_ms else f_ms) if host_ms < f_ms else None,
           "gpu_side_note": "frame_ms is HIP-event time over queued frames; host_ms_* is what issuing a frame costs the calling thread (no wait): where it is "
                            "the smaller number the frame time IS the GPU side (gpu_side_ms), else the frame is bound by the host",
           "pose_ms": p_ms, "skin_ms": s_ms, "scene_frames_per_s": 1e3 / (min(f_ms, f2_ms) if f2_ms else f_ms),
           "skinned_vertices_per_s": total / ((min(f_ms, f2_ms) if f2_ms else f_ms) * 1e-3),
           "skin_roofline": {"bound": "hbm", "kernel": "lbs_skin_batch", "algorithmic_bytes_per_launch": total * 100,
                             "achieved": total * 100 / (s_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": total * 100 / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launch_period_us": s_ms * 1e3},
           "parity": {"characters_checked": [0, n_chars - 1], "frames_in_lock_step": n_par, "bit_exact": exact,
                      "note": "quaternion tracks: palettes and skinned positions bit-identical to the oracle"}}
    for an, mid, *_ in chars:
        an.free()
        ctx.mesh_free(mid)
    for b in frees:
        b.free()
    return rec


def extras(ctx) -> dict:
    """The sub-records.  One that fails (a parity check raises SystemExit) is reported as {"error": ...}: it never costs the others."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import anim_cases as cases
    from fyrox_amd import synth
    streams = ctx.get_option("lbs.streams")
    out = {}

    def guarded(key, fn):
        ctx.set_option("lbs.streams", 1)
        try:
            out[key] = fn()
        except BaseException as e:     # noqa: BLE001
            out[key] = {"error": repr(e)}
            for k, v in (("lbs.exact", 1), ("lbs.timing", 0), ("anim.overlap", 0), ("lbs.crowd_lean", 0)):
                ctx.set_option(k, v)
        ctx.set_option("lbs.streams", streams)

    def c2():
        rig2 = synth.make_rig(64, synth.SEED_BASE + 2)
        td, tgt = synth.make_clip(64, synth.SEED_BASE + 2, 0)
        sc = cases.Scenario("c2", rig2, [td], [cases.AnimSpec(0, tgt)], None, n_frames=20)
        return _chain_record(ctx, "C2: one character, 50k verts / 64 bones / 1 clip (AnimationPlayer -> palette -> LBS)",
                             sc, synth.make_mesh(50_000, 64, synth.SEED_BASE + 2), 1, 400, False, [0])

    guarded("c2", c2)
    guarded("c3", lambda: _c3_record(ctx))
    if isinstance(out.get("c3"), dict) and "fused" in out["c3"]:
        out["c3_fused"] = out["c3"].pop("fused")
    guarded("c3_root_motion", lambda: _c3_record(ctx, frames=200, parity_instances=(0, 999), fused=False, root_motion=True))
    guarded("c5", lambda: _chain_record(ctx, "C5: Machine 4-clip blend tree -> palette -> 100k-vert LBS",
                                        cases.c5_blend_tree(n_bones=64), synth.make_mesh(100_000, 64, synth.SEED_BASE + 5), 1, 400, False, [0]))
    ctx.set_option("lbs.streams", streams)
    guarded("c3_random_bones", lambda: _c3_random_record(ctx))
    for key, fn in (("scene_64x4", lambda: _scene_record(ctx, 64, 4, 20_000, 1_000_000)), ("scene_256x1", lambda: _scene_record(ctx, 256, 1, 5_000, 2_000_000)),
                    ("vertex_buffer", lambda: _vertex_buffer_record(ctx))):
        try:
            out[key] = fn()
        except BaseException as e:     # noqa: BLE001
            out[key] = {"error": repr(e)}
    out["host_control_plane"] = _host_control_plane_record()
    return out


def _host_control_plane_record(frames: int = 200) -> dict:
    """The host half of the pose path ALONE, on this box's cores (a control-only context: no GPU involved): microseconds
    per frame to plan C3's 1000 machines (with the fold-program memo and from scratch) and a scene of 256 single-instance
    characters -- what tools/bench_planner.py and tools/bench_scene_planner.py measure.  Never fails the line."""
    try:
        import fyrox_amd
        from fyrox_amd import anim as A, synth
        c = fyrox_amd.Context(control_only=True)
        try:
            seed = synth.SEED_BASE + 3
            rig = synth.make_rig(64, seed)
            A.create_rig(c, 1, rig)
            tgts = []
            for k in range(4):
                td, tgt = synth.make_clip(64, seed, clip=k)
                A.upload_tracks_data(c, 10 + k, td)
                tgts.append(tgt)

            def character(animator_id, n_instances):
                an = A.Animator(c, animator_id, 1, rig, n_instances)
                for k in range(4):
                    an.add_animation(10 + k, tgts[k], time_slice=(0.0, 1.0), speed=[1.0, 0.8, 1.3, -0.7][k])
                an.set_machine(synth.make_c5_machine())
                return an

            crowd = character(50, 1000)
            for i in range(1000):
                for k in range(4):
                    crowd.set_time_position(k, (i * 0.37 + k * 0.11) % 1.0, instance=i)
            plan, setloop, splan = c._l.fyx_animator_plan, c._l.fyx_animation_set_loop, c._l.fyx_scene_plan
            n, dt = ctypes.c_uint32(), ctypes.c_float(1 / 60)

            def best(fn):
                for _ in range(30):
                    fn()
                b = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(frames):
                        fn()
                    b = min(b, (time.perf_counter() - t0) / frames)
                return b * 1e6

            memo = best(lambda: plan(c._h, 50, 1, dt, None, None, None, None, 0, ctypes.byref(n)))

            def scratch_frame():
                setloop(c._h, 50, 0, 0xFFFFFFFF, 1)       # the value it has: only invalidates the memo
                plan(c._h, 50, 1, dt, None, None, None, None, 0, ctypes.byref(n))
            scratch = best(scratch_frame)
            ids = np.arange(100, 356, dtype=np.uint64)
            for k in ids:
                character(int(k), 1)
            p = ids.ctypes.data_as(ctypes.c_void_p)
            scene = best(lambda: splan(c._h, p, len(ids), dt))
            return {"workload": "fyx_animator_plan over 1000 instances of the C5 machine (C3's control plane); fyx_scene_plan over 256 "
                                "single-instance characters; control-only context, calling thread only",
                    "c3_plan_us_with_memo": memo, "c3_plan_us_from_scratch": scratch, "scene_256x1_plan_us": scene,
                    "host_threads_available": os.cpu_count()}
        finally:
            c.close()
    except Exception as e:     # noqa: BLE001
        return {"error": repr(e)}


# ---- main ---------------------------------------------------------------------------------------------------------------

def launcher_command(n_gpus: int, argv: list) -> list:
    """What `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) re-executes itself as: the driver's own launch line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def run_under_launcher(cmd: list, timeout_s: float, runner=None):
    """The plain `python bench.py --gpus N` road: run the launcher's command as a child, give it `timeout_s`, and say whether it
    produced the line.  Returns (line or None, reason).  `runner` (tests): a stand-in for subprocess.run."""
    import subprocess
    run = runner or subprocess.run
    try:
        cp = run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return None, f"torch.distributed.run did not finish within {timeout_s:.0f} s"
    except Exception as e:     # noqa: BLE001
        return None, f"torch.distributed.run could not be started: {e!r}"
    if cp.stderr:
        sys.stderr.write(cp.stderr[-4000:])
    lines = [ln for ln in (cp.stdout or "").splitlines() if ln.startswith("{") and '"metric"' in ln]
    if cp.returncode == 0 and lines:
        return lines[-1], "ok"
    return None, f"torch.distributed.run exited with {cp.returncode} and {'no' if not lines else 'a'} line; stderr tail: {(cp.stderr or '')[-300:]!r}"


def main_one_process(args, reason: str) -> None:
    """N GPUs from ONE process (the engine's shape: one process, one update thread per ... here one host thread per GPU so that the launch
    calls of 16-us kernels do not queue behind one another): no launcher, no torch process group.  Contexts on devices 0 .. N - 1 (test
    hook FYX_BENCH_DEVICE: all on that device, which fyx_comm_init_all refuses -- the line then says so in comm_error and carries the
    compute legs only).  Legs: strong_value (ONE mesh cut by vertex range, compute only: `value` unless --scaling weak), weak_value (every GPU
    its own mesh), value_with_gather (strong + fyx_allgather_skinned_all / _padded_all, each under a watchdog)."""
    import threading
    import fyrox_amd
    from fyrox_amd import sharding, synth
    n = args.gpus
    forced = os.environ.get("FYX_BENCH_DEVICE")
    devices = [int(forced)] * n if forced else list(range(n))
    ctxs = [fyrox_amd.Context(d) for d in devices]
    for c in ctxs:
        for kv in args.opt:
            k, v = kv.split("=")
            c.set_option(k, int(v))
    nv, nb = args.verts, args.bones
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 4, coherent=not args.random_bones)
    pal = synth.make_palette(nb, synth.SEED_BASE + 4)
    sets = max(2, min(args.sets, 4))
    fn = ctxs[0]._l.fyx_lbs_skin_device
    weak_calls, strong_calls, strong_bufs, strong_bufs_padded, strong_calls_padded = [], [], [], [], []
    shard_p = sharding.vertex_range_padded(nv, 0, n)[2]
    for g, c in enumerate(ctxs):
        d_pal = c.to_device(pal)
        calls = []
        for s_ in range(sets):
            c.mesh_upload_soa(100 + s_, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
            o = tuple(c.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64]))
            calls.append(partial(fn, c._h, ctypes.c_uint64(100 + s_), ctypes.c_void_p(d_pal.ptr), ctypes.c_uint32(nb), ctypes.c_uint32(1),
                                 ctypes.c_void_p(o[0].ptr), ctypes.c_void_p(o[1].ptr), ctypes.c_void_p(o[2].ptr)))
            if g == 0 and s_ == 0:
                first_out = o
        weak_calls.append(calls)
        # strong: this GPU's shard of the ONE mesh, written in place into full-size buffers (ragged cut) / padded buffers (equal cut)
        b, e = sharding.vertex_range_native(nv, g, n)
        c.mesh_upload_soa(200, mesh.pos[b:e], mesh.weights[b:e], mesh.indices[b:e], mesh.normal[b:e], mesh.tangent[b:e])
        full = tuple(c.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64]))
        strong_bufs.append(full)
        strong_calls.append(partial(fn, c._h, ctypes.c_uint64(200), ctypes.c_void_p(d_pal.ptr), ctypes.c_uint32(nb), ctypes.c_uint32(1),
                                    ctypes.c_void_p(full[0].ptr + 12 * b), ctypes.c_void_p(full[1].ptr + 12 * b), ctypes.c_void_p(full[2].ptr + 16 * b))
                            if e > b else (lambda: 0))
        b2, e2, _ = sharding.vertex_range_padded(nv, g, n)
        c.mesh_upload_soa(201, mesh.pos[b2:e2], mesh.weights[b2:e2], mesh.indices[b2:e2], mesh.normal[b2:e2], mesh.tangent[b2:e2])
        fullp = tuple(c.malloc_streams([n * shard_p * 12 + 64, n * shard_p * 12 + 64, n * shard_p * 16 + 64]))
        strong_bufs_padded.append(fullp)
        strong_calls_padded.append(partial(fn, c._h, ctypes.c_uint64(201), ctypes.c_void_p(d_pal.ptr), ctypes.c_uint32(nb), ctypes.c_uint32(1),
                                           ctypes.c_void_p(fullp[0].ptr + 12 * b2), ctypes.c_void_p(fullp[1].ptr + 12 * b2), ctypes.c_void_p(fullp[2].ptr + 16 * b2))
                                   if e2 > b2 else (lambda: 0))
    parity = None
    if not args.no_check:
        weak_calls[0][0]()
        ctxs[0].sync()
        parity = lbs_parity(ctxs[0], mesh, pal, tuple(_Ptr(b.ptr) for b in first_out), 20_000)
        if not parity["bit_exact"]:
            raise SystemExit("one-process bench: GPU 0's output differs from the oracle")

    def timed(per_gpu_step, steps, warmup, after=None):
        """Every GPU's `steps` launches issued by a thread of its own between a sync of all contexts before and after; `after(i)`
        (the exchange) is called from THIS thread once per step when given -- then the launches are issued from this thread too,
        step by step, as the engine's single update thread would."""
        def run(count):
            if after is not None:
                for i in range(count):
                    for g in range(n):
                        per_gpu_step(g, i)
                    after(i)
                return
            ths = [threading.Thread(target=lambda g=g: [per_gpu_step(g, i) for i in range(count)]) for g in range(n)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        run(warmup)
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        run(steps)
        for c in ctxs:
            c.sync()
        return time.perf_counter() - t0

    def chk(rc):
        if rc:
            raise RuntimeError(ctxs[0]._l.fyx_last_error(ctxs[0]._h).decode())

    w = timed(lambda g, i: chk(weak_calls[g][i % sets]()), args.steps, args.warmup)
    s_c = timed(lambda g, i: chk(strong_calls[g]()), args.steps, args.warmup)
    weak_v, weak_ms, strong_v, strong_ms = float(nv) * n * args.steps / w, w * 1e3 / args.steps, float(nv) * args.steps / s_c, s_c * 1e3 / args.steps
    as_weak = args.scaling == "weak"
    out = {"metric": "skinned vertices/sec at 1M verts/256 bones; achieved HBM GB/s vs peak",
           "value": weak_v if as_weak else strong_v, "unit": "vertices/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": weak_ms if as_weak else strong_ms, "higher_is_better": True, "scaling": "weak" if as_weak else "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic", "timed_steps": args.steps, "repeats": 1,
           "config": {"workload": (f"C4: {nv} verts / {nb} bones per GPU, 4-influence LBS of position+normal+tangent, {sets} rotating buffer sets; `value` is WEAK "
                                   f"scaling over the {n} GPUs (every GPU its own mesh, no collective); strong_value = BASELINE config 4 as written") if as_weak else
                                  (f"C4 as written: {nv} verts / {nb} bones cut by contiguous vertex range over {n} GPU(s), palette replicated, every GPU writes its shard "
                                   "in place; `value` is this STRONG-scaling job, compute only; value_with_gather = the same with the RCCL exchange (fastest form); "
                                   "weak_value = every GPU its own 1 M-vertex mesh"),
                      "process_group": f"none: ONE process drives the {n} GPUs (fyx_comm_init_all / fyx_allgather_skinned_all), one host thread per GPU; {reason}",
                      "devices": devices, "sharding": "contiguous vertex range per GPU, palette replicated", "n_ranks": n, "sets": sets},
           "roofline": None, "parity": parity,
           "strong_value": strong_v, "strong_ms_per_step": strong_ms, "weak_value": weak_v, "weak_ms_per_step": weak_ms,
           "value_with_gather": None, "value_with_gather_form": None, "comm_error": None,
           "timing_note": "host clock between fyx_sync of every context before and after the launches (no torch, no events across devices)",
           "box": {"at_start": box_facts(devices[0])}}
    # the exchange, last and under a watchdog: RCCL with more than one rank runs here for the first time
    try:
        fyrox_amd.Context.comm_init_all(ctxs)
        have_comm = True
    except Exception as e:     # noqa: BLE001
        have_comm, out["comm_error"] = False, repr(e)
    if have_comm:
        legs = {}
        for form, key in ((0, "broadcasts"), (1, "send_recv"), (2, "all_gather_padded")):
            done = threading.Event()

            def give_up(key=key):
                if not done.is_set():
                    legs[key] = {"value": None, "note": f"the exchange did not finish within {EXCHANGE_TIMEOUT_S} s"}
                    out["exchange_legs"] = legs
                    finish(out, args.full_record)
                    os._exit(0)
            wd = threading.Timer(EXCHANGE_TIMEOUT_S, give_up)
            wd.daemon = True
            wd.start()
            try:
                if form == 2:
                    bufs, calls = strong_bufs_padded, strong_calls_padded
                    gather = lambda i: fyrox_amd.Context.allgather_skinned_padded_all(ctxs, nv, n * shard_p, [b[0].ptr for b in bufs], [b[1].ptr for b in bufs], [b[2].ptr for b in bufs])   # noqa: E731
                else:
                    ctxs[0].set_option("comm.form", form)
                    bufs, calls = strong_bufs, strong_calls
                    gather = lambda i: fyrox_amd.Context.allgather_skinned_all(ctxs, nv, [b[0].ptr for b in bufs], [b[1].ptr for b in bufs], [b[2].ptr for b in bufs])   # noqa: E731
                t = timed(lambda g, i: chk(calls[g]()), args.steps, min(args.warmup, 20), after=gather)
                ok = None
                if not args.no_check:      # the LAST GPU holds the whole mesh: its head (another GPU's shard) against the oracle
                    ok = lbs_parity(ctxs[-1], mesh, pal, tuple(_Ptr(b.ptr) for b in bufs[-1]), 20_000)["bit_exact"]
                legs[key] = {"value": float(nv) * args.steps / t, "ms_per_step": t * 1e3 / args.steps, "gathered_equals_oracle": ok}
                if ok is not False and (out["value_with_gather"] is None or legs[key]["value"] > out["value_with_gather"]):
                    out["value_with_gather"], out["value_with_gather_form"] = legs[key]["value"], key
            except Exception as e:     # noqa: BLE001
                legs[key] = {"value": None, "note": f"the exchange failed: {e!r}"}
                done.set()
                wd.cancel()
                break
            done.set()
            wd.cancel()
        ctxs[0].set_option("comm.form", 0)
        out["exchange_legs"] = legs
    out["box"]["at_end"] = box_facts(devices[0])
    finish(out, args.full_record)
    for c in ctxs:
        c.close()


def check_world(n_gpus: int, world: int) -> None:
    """One rank per GPU or no line at all: `n_gpus` in the line is the number of ranks that ran."""
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: the line would carry the wrong n_gpus")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.extras_only:
        if args.one_process:
            return main_one_process(args, "asked for with --one-process")
        # `python bench.py --gpus N` launched plainly: one rank per GPU is what the line promises, so this process runs the launcher the
        # driver would have used (same module, same arguments) as a child -- and when that road fails or does not finish (the launcher,
        # torch's rendezvous, a process group that never forms) takes the second one: all N GPUs from this process
        cmd = launcher_command(args.gpus, sys.argv[1:])
        print(f"# bench.py --gpus {args.gpus} without a launcher: running {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
        line, reason = run_under_launcher(cmd, args.launcher_timeout)
        if line is not None:
            emit(line)
            return None
        print(f"# {reason}; falling back to --one-process", file=sys.stderr, flush=True)
        return main_one_process(args, f"fallback: {reason}")
    check_world(args.gpus, world)
    if args.scaling is None:
        args.scaling = "weak" if world == 1 else "strong"
    run_weak = args.scaling == "weak" or world > 1        # N > 1: both scalings are measured, --scaling says which one is `value`

    if args.extras_only:
        # The other BASELINE configs and the scene tick, in a process of their own: their pipelined frames are bound by the host's
        # stream / event calls, and in the process that has run the headline legs those calls are ~15 % slower (bisected to "after the
        # headline's set-up and first launches"; not to a cause -- neither torch's default-stream copies, nor live events, nor the
        # oracle's OpenMP burst, each ruled out by experiment).  The same functions, the same library, a fresh HIP runtime.
        import fyrox_amd
        ctx = fyrox_amd.Context(local_rank)
        for kv in args.opt:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        emit(json.dumps(extras(ctx)))
        ctx.close()
        return

    import torch
    import fyrox_amd
    from fyrox_amd import sharding, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # FYX_BENCH_DEVICE: test hook -- every rank on this device (exercises the N > 1 code on a one-GPU box; RCCL refuses
    # two ranks on one GPU, so it also exercises the fall-backs below)
    if os.environ.get("FYX_BENCH_DEVICE"):
        local_rank = int(os.environ["FYX_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist, dist_backend, dist_cuda = None, None, True
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the process group only carries barriers, the max over ranks and the 128-byte unique id; RCCL when it comes up,
        # gloo otherwise (the decision is each rank's own: a failing RCCL start-up fails on every rank)
        try:
            if os.environ.get("FYX_BENCH_DEVICE"):
                raise RuntimeError("ranks share a device")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist_backend = "nccl"
        except Exception as e:     # noqa: BLE001
            print(f"# rank {rank}: process group on RCCL failed ({e!r}); using gloo for the control collectives", file=sys.stderr)
            try:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            except Exception as e2:     # noqa: BLE001
                # no process group at all under the launcher: rank 0 takes the launcher-free road over all the GPUs, the others leave
                if rank != 0:
                    return None
                return main_one_process(args, f"fallback under the launcher: no torch process group ({e2!r})")
            dist_backend, dist_cuda = "gloo", False

    ctx = fyrox_amd.Context(local_rank)    # owns its launch streams; torch is only used for barriers and buffers
    box = {"at_start": box_facts(local_rank)} if rank == 0 else {}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        box["hip_device"] = {"name": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None), "multi_processor_count": pr.multi_processor_count,
                             "total_memory": pr.total_memory, "clock_rate_khz": getattr(pr, "clock_rate", None),
                             "memory_clock_rate_khz": getattr(pr, "memory_clock_rate", None), "l2_cache_size": getattr(pr, "L2_cache_size", None)}
    except Exception as e:     # noqa: BLE001
        box["hip_device"] = {"error": repr(e)}
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    opt_keys = ("lbs.blocks_per_cu", "lbs.exact", "lbs.streams", "lbs.dyn", "lbs.crowd", "anim.inline_ctrl", "anim.ctrl_upload", "anim.update_lean", "anim.update_pack", "anim.one_launch",
                "streams.priority", "comm.form")
    opts = {k: ctx.get_option(k) for k in opt_keys}
    n_ranks_rccl, comm_error = None, None
    if world > 1:      # the library's own communicator (fyx_comm_init): rank 0's unique id travels over the process group
        ok = 1
        try:
            uid = [ctx.comm_unique_id() if rank == 0 else None]
        except Exception as e:     # noqa: BLE001
            uid, ok, comm_error = [None], 0, repr(e)
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            ok = 0
        else:
            try:
                ctx.comm_init(uid[0], rank, world)
                n_ranks_rccl = ctx.comm_info()[1]
            except Exception as e:     # noqa: BLE001
                ok, comm_error = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist_cuda else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)       # all ranks take the same path
        if int(flag[0]) == 0:
            n_ranks_rccl = None
            comm_error = comm_error or "another rank could not join the communicator"
            print(f"# rank {rank}: fyx_comm_init failed ({comm_error}); the exchange leg is skipped", file=sys.stderr)
    force_exchange = world == 1 and bool(os.environ.get("FYX_BENCH_FORCE_EXCHANGE"))   # test hook: single-rank communicator
    if force_exchange:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
        n_ranks_rccl = ctx.comm_info()[1]
    have_comm = n_ranks_rccl is not None

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if dist_cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def timed_regions(step, steps, warmup, first_index=0):
        """One timed region = `repeats` back-to-back passes of `steps` launches between barrier + device sync (no sync
        inside); `repeats` is 1 when `steps` launches already last MIN_REGION_MS, otherwise what makes the region that
        long (from an untimed pilot pass), so that the fixed cost of opening and closing a region (~50 us) does not
        decide a number quoted per step.  Returns (repeats, per-region wall seconds (max over ranks), GPU ms)."""
        for i in range(warmup):
            step(first_index + i)
        n = first_index + warmup

        def region(count):
            nonlocal n
            barrier()
            t0 = time.perf_counter()
            ctx.timer_begin()
            for i in range(count):
                step(n + i)
            g = ctx.timer_end()
            barrier()
            n += count
            return max_over_ranks(time.perf_counter() - t0), max_over_ranks(g)

        pilot, _ = region(steps)
        repeats = max(1, min(args.max_repeats, int(np.ceil(MIN_REGION_MS * 1e-3 / max(pilot, 1e-6)))))
        walls, gpus = [], []
        for _ in range(5 if repeats > 1 else 1):
            w, g = region(steps * repeats)
            walls.append(w)
            gpus.append(g)
        return repeats, walls, gpus


    seed = synth.SEED_BASE + 4
    pal = synth.make_palette(args.bones, seed)
    d_pal = torch.from_numpy(pal).cuda()
    fn = ctx._l.fyx_lbs_skin_device

    # ---- headline: every rank skins its own 1 M-vertex shard (weak); N = 1: THE C4 workload ----------------------------
    shard = sharding.vertex_range(world * args.verts, rank, world)
    mesh = synth.make_mesh(args.verts, args.bones, seed + 1000 * rank, coherent=not args.random_bones)
    nv = mesh.n_verts
    outs = []
    for s in range(args.sets):
        ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        outs.append(output_streams(ctx, nv))       # every output stream its own allocation, made by the library: the same on every box
    # One foreign call per step with the ctypes arguments converted once: ~1 us of Python per launch.
    calls = [partial(fn, ctx._h, ctypes.c_uint64(s), ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones),
                     ctypes.c_uint32(1), ctypes.c_void_p(o[0].data_ptr()), ctypes.c_void_p(o[1].data_ptr()),
                     ctypes.c_void_p(o[2].data_ptr())) for s, o in enumerate(outs)]
    n_sets = args.sets

    def step(i: int):
        rc = calls[i % n_sets]()
        if rc:
            ctx._check(rc)

    parity = None
    if not args.no_check and rank == 0:
        for o in outs[0]:
            zero(ctx, o)
        step(0)
        ctx.sync()
        parity = lbs_parity(ctx, mesh, pal, outs[0], nv)        # every vertex of the launch (the oracle on 8 threads: ~0.2 s for 1 M)
        if parity["max_rel_err"] > 1e-5:
            raise SystemExit(f"parity check failed before timing: max rel err {parity['max_rel_err']:.3e} > 1e-5")

    strong = None
    weak_regions = None
    if run_weak:
        weak_regions = timed_regions(step, args.steps, args.warmup)
        repeats, walls, gpus = weak_regions
    # ---- the kernel alone: launches serialized on ONE stream (HIP-event average per launch == rocprofv3 kernel duration) --
    n_ser = max(500, min(args.steps, 2000))
    ctx.set_option("lbs.streams", 1)
    for i in range(50):
        step(i)
    ser, ker = [], []
    for _ in range(3):
        ctx.sync()
        ctx.timer_begin()
        for i in range(n_ser):
            step(i)
        ser.append(ctx.timer_end() * 1e3 / n_ser)
    ctx.set_option("lbs.timing", 1)      # every launch carries its own start / stop events (the dispatch's timestamps)
    ctx.kernel_time()
    for _ in range(3):
        for i in range(n_ser):
            step(i)
        us, n = ctx.kernel_time()
        ker.append(us / max(n, 1))
    ctx.set_option("lbs.timing", 0)
    # The same measurement on FRESH allocations of the eight output sets (hipMalloc through the library, the first sets still held, so
    # the addresses differ): where the pages of the outputs land moves a lone launch by several per cent, and one allocation is one
    # sample of that.  roofline.kernel_us is the median over all allocations of this process.
    alloc_us = [float(np.median(ker))]
    if rank == 0 and world == 1:
        held = []
        for rep in range(3):
            fresh = [tuple(ctx.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64])) for _ in range(n_sets)]
            held.append(fresh)
            fcalls = [partial(fn, ctx._h, ctypes.c_uint64(s_), ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones), ctypes.c_uint32(1),
                              ctypes.c_void_p(o[0].ptr), ctypes.c_void_p(o[1].ptr), ctypes.c_void_p(o[2].ptr)) for s_, o in enumerate(fresh)]
            ctx.set_option("lbs.streams", 1)
            for i in range(50):
                fcalls[i % n_sets]()
            ctx.set_option("lbs.timing", 1)
            ctx.kernel_time()
            for i in range(n_ser):
                fcalls[i % n_sets]()
            us, n = ctx.kernel_time()
            ctx.set_option("lbs.timing", 0)
            alloc_us.append(us / max(n, 1))
        for fresh in held:
            for o in fresh:
                for b_ in o:
                    b_.free()
        ctx.set_option("lbs.streams", opts["lbs.streams"])
    # The same launch against the NUMBER of rotating 100 MB sets (round 5: tools/exp/r05_sets_sweep.py).  The rotation exists to keep the
    # 256 MiB Infinity Cache out of the number; how many sets it takes also decides how much memory the launches walk over, and the kernel's
    # time steps with that footprint (one set -- what an engine's frame loop does -- is the fastest; 2 - 6 sets are flat; past ~700 MB
    # it climbs again and levels off beyond ~1.2 GB: address translation, not the cache).  roofline.frac stays on --sets (8: comparable
    # with rounds 1 - 4); SURVEY 8(d) prescribes ">= 6 sets".
    by_sets = None
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            extra_outs, more = [], list(calls)
            for s_ in range(n_sets, 16):
                ctx.mesh_upload_soa(5000 + s_, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
                o = tuple(ctx.malloc_streams([nv * 12 + 64, nv * 12 + 64, nv * 16 + 64]))
                extra_outs.append(o)
                more.append(partial(fn, ctx._h, ctypes.c_uint64(5000 + s_), ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones), ctypes.c_uint32(1),
                                    ctypes.c_void_p(o[0].ptr), ctypes.c_void_p(o[1].ptr), ctypes.c_void_p(o[2].ptr)))
            ctx.set_option("lbs.streams", 1)
            by_sets = {}
            for k in sorted({1, 2, 4, 6, n_sets, 12, 16}):
                if k > len(more):
                    continue
                for i in range(40):
                    more[i % k]()
                ctx.set_option("lbs.timing", 1)
                ctx.kernel_time()
                for i in range(600):
                    more[i % k]()
                us, n = ctx.kernel_time()
                ctx.set_option("lbs.timing", 0)
                by_sets[str(k)] = {"kernel_us": us / max(n, 1), "frac": BYTES_PER_VERTEX * nv / (us / max(n, 1) * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                   "footprint_MB": k * BYTES_PER_VERTEX * nv / 1e6}
            # Inputs and outputs rotated SEPARATELY (round 6): what the launch pays for is the number of OUTPUT sets -- while they fit the
            # 256 MiB Infinity Cache (<= 6 x 40 MB) the memory-side cache absorbs the launch's stores and HBM carries the 60 MB of reads
            # alone; past it every store reaches HBM.  The number of input sets does not matter beyond 2 (profiles/r06_lone_launch/).
            if len(more) >= 16:
                all_in = [ctypes.c_uint64(s_) for s_ in range(n_sets)] + [ctypes.c_uint64(5000 + s_) for s_ in range(n_sets, 16)]
                all_out = [tuple(ctypes.c_void_p(t_.data_ptr()) for t_ in o) for o in outs] + [tuple(ctypes.c_void_p(b_.ptr) for b_ in o) for o in extra_outs]

                def split_case(n_in, n_out):
                    cs = [partial(fn, ctx._h, all_in[i % n_in], ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones), ctypes.c_uint32(1),
                                  *all_out[i % n_out]) for i in range(16)]
                    for i in range(48):
                        cs[i % 16]()
                    ctx.set_option("lbs.timing", 1)
                    ctx.kernel_time()
                    for i in range(608):
                        cs[i % 16]()
                    us_, n_ = ctx.kernel_time()
                    ctx.set_option("lbs.timing", 0)
                    k_ = us_ / max(n_, 1)
                    return {"kernel_us": k_, "frac": BYTES_PER_VERTEX * nv / (k_ * 1e-6) / 1e9 / HBM_PEAK_GBPS}
                by_sets["inputs_vs_outputs"] = {f"in{i_}_out{o_}": split_case(i_, o_) for i_, o_ in ((16, 1), (16, 6), (16, 8), (16, 16), (1, 16), (1, 1))}
            for o in extra_outs:
                for b_ in o:
                    b_.free()
            for s_ in range(n_sets, 16):
                ctx.mesh_free(5000 + s_)
        except Exception as e:     # noqa: BLE001
            by_sets = {"error": repr(e)}
            ctx.set_option("lbs.timing", 0)
        ctx.set_option("lbs.streams", opts["lbs.streams"])
    kernel_us = max_over_ranks(float(np.median(alloc_us)))     # the kernel's own duration: median over the allocations, each averaged over n_ser launches
    period_us = max_over_ranks(float(np.median(ser)))          # launch to launch on one stream (adds the dependent-launch gap)
    ctx.set_option("lbs.streams", opts["lbs.streams"])
    # ---- position only: what the reference's CPU loop computes (mesh/mod.rs:501-522): 32 B read + 12 B written per vertex ----
    pos_only = None
    if rank == 0:
        pcalls = [partial(fn, ctx._h, ctypes.c_uint64(s), ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones),
                          ctypes.c_uint32(1), ctypes.c_void_p(o[0].data_ptr()), ctypes.c_void_p(None), ctypes.c_void_p(None))
                  for s, o in enumerate(outs)]
        ctx.set_option("lbs.streams", 1)
        for i in range(50):
            pcalls[i % n_sets]()
        ctx.set_option("lbs.timing", 1)
        ctx.kernel_time()
        for i in range(n_ser):
            pcalls[i % n_sets]()
        us, n = ctx.kernel_time()
        ctx.set_option("lbs.timing", 0)
        ctx.set_option("lbs.streams", opts["lbs.streams"])
        for i in range(50):
            pcalls[i % n_sets]()
        ctx.sync()
        ctx.timer_begin()
        for i in range(n_ser):
            pcalls[i % n_sets]()
        ov = ctx.timer_end() * 1e3 / n_ser
        pk = us / max(n, 1)
        pos_only = {"workload": "C4, position stream only (the reference's CPU loop): 44 B/vertex", "algorithmic_bytes_per_launch": 44 * nv,
                    "kernel_us": pk, "frac": 44 * nv / (pk * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                    "overlapped_us": ov, "overlapped_frac": 44 * nv / (ov * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                    "vertices_per_s_overlapped": nv / (ov * 1e-6)}
    # ---- the worst case SURVEY 8(d) defines: FULLY RANDOM bone indices (a wave's 64 lanes gather 64 unrelated palette rows from LDS
    # per influence instead of mostly the same few), same launch, same buffers, same run ----
    random_rec = None
    if rank == 0 and world == 1 and not args.random_bones and not args.no_extras:
        rmesh = synth.make_mesh(args.verts, args.bones, seed, coherent=False)
        for s_ in range(n_sets):
            ctx.mesh_upload_soa(300 + s_, rmesh.pos, rmesh.weights, rmesh.indices, rmesh.normal, rmesh.tangent)
        rcalls = [partial(fn, ctx._h, ctypes.c_uint64(300 + s_), ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones), ctypes.c_uint32(1),
                          ctypes.c_void_p(o[0].data_ptr()), ctypes.c_void_p(o[1].data_ptr()), ctypes.c_void_p(o[2].data_ptr())) for s_, o in enumerate(outs)]
        for o in outs[0]:
            zero(ctx, o)
        rcalls[0]()
        ctx.sync()
        rpar = lbs_parity(ctx, rmesh, pal, outs[0], nv)
        if not rpar["bit_exact"]:
            raise SystemExit("random bone indices: the launch differs from the oracle")
        ctx.set_option("lbs.streams", 1)
        for i in range(50):
            rcalls[i % n_sets]()
        ctx.set_option("lbs.timing", 1)
        ctx.kernel_time()
        rk = []
        for _ in range(3):
            for i in range(n_ser):
                rcalls[i % n_sets]()
            us, n = ctx.kernel_time()
            rk.append(us / max(n, 1))
        ctx.set_option("lbs.timing", 0)
        ctx.set_option("lbs.streams", opts["lbs.streams"])
        for i in range(50):
            rcalls[i % n_sets]()
        ctx.sync()
        ctx.timer_begin()
        for i in range(n_ser):
            rcalls[i % n_sets]()
        rov = ctx.timer_end() * 1e3 / n_ser
        rkm = float(np.median(rk))
        random_rec = {"workload": f"C4 with FULLY RANDOM bone indices (SURVEY 8(d) worst case): {nv} verts / {args.bones} bones, same launch and buffers",
                      "kernel_us": rkm, "frac": BYTES_PER_VERTEX * nv / (rkm * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                      "kernel_us_coherent_same_run": kernel_us, "slowdown_vs_coherent": rkm / kernel_us - 1.0,
                      "overlapped_us": rov, "overlapped_frac": BYTES_PER_VERTEX * nv / (rov * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                      "value": nv / (rov * 1e-6), "unit": "vertices/s (launches overlapped on the launch streams, as the headline)",
                      "parity": rpar}
        for s_ in range(n_sets):
            ctx.mesh_free(300 + s_)
    # ---- no-math copy of the same bytes (60 MB in, 40 MB out per launch), same run: what a 100 MB launch can do here ----
    copy_us = None
    if rank == 0:
        units = 1_250_000
        srcs = [torch.full((units * 12,), float(s), dtype=torch.float32, device="cuda") for s in range(n_sets)]
        dsts = [torch.empty(units * 8, dtype=torch.float32, device="cuda") for _ in range(n_sets)]
        for i in range(40):
            ctx.calib_stream_copy(srcs[i % n_sets].data_ptr(), dsts[i % n_sets].data_ptr(), units)
        cs = []
        for _ in range(3):
            ctx.sync()
            ctx.timer_begin()
            for i in range(500):
                ctx.calib_stream_copy(srcs[i % n_sets].data_ptr(), dsts[i % n_sets].data_ptr(), units)
            cs.append(ctx.timer_end() * 1e3 / 500)
        copy_us = float(np.median(cs))
        del srcs, dsts

    # ---- BASELINE config 4 as written: the 1 M-vertex mesh cut by vertex range over the ranks ---------------------------
    if world > 1 or args.scaling == "strong" or force_exchange:
        full = synth.make_mesh(args.verts, args.bones, seed, coherent=not args.random_bones)     # same mesh on every rank
        b, e = sharding.vertex_range_native(full.n_verts, rank, world)
        sets2 = min(n_sets, 4)
        alls = []
        for s in range(sets2):
            ctx.mesh_upload_soa(100 + s, full.pos[b:e], full.weights[b:e], full.indices[b:e], full.normal[b:e], full.tangent[b:e])
            alls.append(output_streams(ctx, full.n_verts))
            for o_ in alls[-1]:
                zero(ctx, o_)
        scalls = [partial(fn, ctx._h, ctypes.c_uint64(100 + s), ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones),
                          ctypes.c_uint32(1), ctypes.c_void_p(o[0].data_ptr() + 12 * b), ctypes.c_void_p(o[1].data_ptr() + 12 * b),
                          ctypes.c_void_p(o[2].data_ptr() + 16 * b)) for s, o in enumerate(alls)]
        gather = ctx._l.fyx_allgather_skinned
        gcalls = [partial(gather, ctx._h, ctypes.c_uint32(full.n_verts), ctypes.c_void_p(o[0].data_ptr()),
                          ctypes.c_void_p(o[1].data_ptr()), ctypes.c_void_p(o[2].data_ptr())) for o in alls]
        # exchange form 2 (one in-place ncclAllGather per stream) wants EQUAL shards: the padded cut, buffers of world * shard vertices
        b2, e2, shard2 = sharding.vertex_range_padded(full.n_verts, rank, world)
        alls2, scalls2, gcalls2 = [], [], []
        for s_ in range(sets2):
            ctx.mesh_upload_soa(150 + s_, full.pos[b2:e2], full.weights[b2:e2], full.indices[b2:e2], full.normal[b2:e2], full.tangent[b2:e2])
            o = output_streams(ctx, world * shard2)
            for o_ in o:
                zero(ctx, o_)
            alls2.append(o)
            scalls2.append(partial(fn, ctx._h, ctypes.c_uint64(150 + s_), ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones), ctypes.c_uint32(1),
                                   ctypes.c_void_p(o[0].data_ptr() + 12 * b2), ctypes.c_void_p(o[1].data_ptr() + 12 * b2), ctypes.c_void_p(o[2].data_ptr() + 16 * b2)))
            gcalls2.append(partial(ctx._l.fyx_allgather_skinned_padded, ctx._h, ctypes.c_uint32(full.n_verts), ctypes.c_uint32(world * shard2),
                                   ctypes.c_void_p(o[0].data_ptr()), ctypes.c_void_p(o[1].data_ptr()), ctypes.c_void_p(o[2].data_ptr())))

        def sstep(i: int, with_gather: bool, padded: bool = False):
            rc = (scalls2 if padded else scalls)[i % sets2]() if (e2 > b2 if padded else e > b) else 0
            if rc:
                ctx._check(rc)
            if with_gather and have_comm:
                rc = (gcalls2 if padded else gcalls)[i % sets2]()
                if rc:
                    ctx._check(rc)

        # compute only first; the exchange (RCCL with more than one rank) runs LAST, under a watchdog, see exchange_leg
        sstep(0, False)
        ctx.sync()
        if rank == 0 and not args.no_check:
            head = lbs_parity(ctx, full, pal, alls[0], 20_000)        # rank 0's own shard starts at vertex 0
            if not head["bit_exact"]:
                raise SystemExit("strong scaling: rank 0's shard differs from the oracle")
        r_c, w_c, g_c = timed_regions(lambda i: sstep(i, False), args.steps, args.warmup)
        sizes = [sharding.vertex_range_native(full.n_verts, r, world) for r in range(world)]
        strong = {"workload": f"C4 as written: {full.n_verts} verts / {args.bones} bones cut by contiguous vertex range over {world} GPU(s), "
                              "palette replicated; every rank writes its shard in place into the full buffers",
                  "scaling": "strong", "n_ranks": n_ranks_rccl if n_ranks_rccl is not None else 1,
                  "shard_vertices": [e_ - b_ for b_, e_ in sizes],
                  "compute_only": {"value": full.n_verts * args.steps * r_c / float(np.median(w_c)), "unit": "vertices/s",
                                   "ms_per_step": float(np.median(w_c)) * 1e3 / (args.steps * r_c), "timed_steps": args.steps * r_c},
                  "with_allgather": {"value": None, "note": "world size 1: nothing to exchange" if world == 1 else
                                     "no communicator (see comm_error): the exchange leg did not run"},
                  "gathered_equals_oracle": None, "comm_error": comm_error}

        def exchange_leg(form: int):
            """Every rank ends up holding the WHOLE skinned mesh (checked on rank 0 against the oracle, head and tail), then the
            timed regions with the exchange in them.  `form`: option comm.form (0 one broadcast per shard, 1 grouped send / recv,
            2 one in-place all-gather per stream over the padded cut).  Returns (record, regions) on rank 0's behalf; all ranks take part."""
            ctx.set_option("comm.form", form)
            padded = form == 2
            bufs = alls2[0] if padded else alls[0]
            for o in bufs:
                zero(ctx, o)
            sstep(0, True, padded)
            ctx.sync()
            ok = None
            if rank == 0 and not args.no_check:
                import oracle
                n_chk = 20_000
                tail = slice(full.n_verts - n_chk, full.n_verts)
                ref = oracle.lbs_skin(full.pos[tail], full.weights[tail], full.indices[tail], pal, full.normal[tail], full.tangent[tail], threads=CHECK_THREADS)
                got = dl(ctx, bufs[0], (full.n_verts - n_chk) * 3, n_chk * 3).reshape(-1, 3)      # (vertex v lies at position v in either cut)
                ok = bool(lbs_parity(ctx, full, pal, bufs, n_chk)["bit_exact"] and np.array_equal(got, ref["pos"]))
            r_g, w_g, g_g = timed_regions(lambda i: sstep(i, True, padded), args.steps, args.warmup)
            rec = {"value": full.n_verts * args.steps * r_g / float(np.median(w_g)), "unit": "vertices/s",
                   "ms_per_step": float(np.median(w_g)) * 1e3 / (args.steps * r_g), "timed_steps": args.steps * r_g,
                   "collective": "fyx_allgather_skinned: one grouped RCCL op per frame (40 B/vertex), " +
                                 ("ragged shards, one ncclBroadcast per (stream, shard)" if form == 0 else
                                  "ragged shards, ncclSend / ncclRecv between every pair of ranks" if form == 1 else
                                  f"equal padded shards of {shard2} vertices, ONE in-place ncclAllGather per stream")}
            return rec, ok, (r_g, w_g, g_g)

        if args.scaling == "strong":
            repeats, walls, gpus = r_c, w_c, g_c

    if rank == 0:
        region = float(np.median(walls))
        per_rank_verts = nv if args.scaling == "weak" else args.verts / world
        timed_steps = args.steps * repeats
        total_verts = (float(world) * nv if args.scaling == "weak" else float(args.verts)) * timed_steps
        value = total_verts / region
        launch_us = float(np.median(gpus)) * 1e3 / timed_steps          # per launch in the timed region (launches overlap)
        bytes_launch = BYTES_PER_VERTEX * nv
        achieved = bytes_launch / (kernel_us * 1e-6) / 1e9
        traffic, traffic_src, traffic_live = None, None, None
        if world == 1 and not args.no_pmc and not args.random_bones:      # measured now, by two short rocprofv3 passes in child processes
            ctx.sync()
            traffic_live = pmc_traffic(nv, args.bones)
            if "error" not in traffic_live:
                traffic, traffic_src = traffic_live["hbm_bytes_per_launch"], traffic_live["source"]
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if traffic is None and os.path.exists(tpath):    # the builder's last PMC run, labelled as such (see profiles/README.md)
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                traffic_src = "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, builder run; not measured in this run)"
            except Exception:
                traffic = None
        kname = "lbs_skin_dyn" if opts["lbs.dyn"] and nv >= 524_288 else "lbs_skin"
        box["after_headline"] = box_facts(local_rank)
        out = {
            "metric": "skinned vertices/sec at 1M verts/256 bones; achieved HBM GB/s vs peak",
            "value": value, "unit": "vertices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": region * 1e3 / timed_steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timed_steps": timed_steps, "repeats": repeats, "timed_regions": len(walls), "region_ms": [w * 1e3 for w in walls],
            "timing_note": f"a timed region = {repeats} back-to-back passes of --steps {args.steps} launches between barrier + sync "
                           f"(no sync inside), so that it lasts >= {MIN_REGION_MS:.0f} ms; median of {len(walls)} regions",
            "config": {"workload": f"C4: {nv} verts / {args.bones} bones per GPU, 4-influence LBS of position+normal+tangent, "
                                   f"{args.sets} rotating 100 MB buffer sets, "
                                   f"{'random' if args.random_bones else 'spatially coherent'} bone indices"
                                   + ("" if world == 1 else f"; `value` is WEAK scaling over the {world} GPUs (every GPU its own 1 M-vertex mesh, no collective); BASELINE "
                                      "config 4 as written (ONE 1 M-vertex mesh cut by vertex range) is strong_value / value_with_gather, config 3 cut by "
                                      "instance range is crowd_value")
                                   if args.scaling == "weak" else strong["workload"] + ("" if world == 1 else
                                   "; `value` is this STRONG-scaling job, compute only; value_with_gather = the same with the RCCL exchange (fastest form); "
                                   "weak_value = every GPU its own 1 M-vertex mesh; crowd_value = config 3 cut by instance range"),
                       "sharding": "contiguous vertex range per GPU, palette replicated",
                       "rank0_vertex_range": list(shard), "n_ranks": n_ranks_rccl if n_ranks_rccl is not None else 1,
                       "process_group": dist_backend, "sets": args.sets,
                       "kernel_options": opts},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src, "traffic_measurement": traffic_live,
                         "kernel": kname, "kernel_us": kernel_us,
                         "kernel_us_min": float(np.min(alloc_us)), "kernel_us_median": float(np.median(alloc_us)), "kernel_us_max": float(np.max(alloc_us)),
                         "kernel_us_per_allocation": alloc_us,
                         "allocations_note": f"{len(alloc_us)} allocations of the {args.sets} output sets in this process (all through fyx_malloc_streams: one "
                                             "device allocation per output stream), each the average of the per-dispatch durations of a few hundred launches; "
                                             "kernel_us and frac are the MEDIAN",
                         "kernel_us_note": "launches serialized on one stream, each with its own start / stop HIP events (hipExtLaunchKernel: the "
                                           f"dispatch's timestamps, what rocprofv3 --kernel-trace reports per dispatch); average of {n_ser} launches",
                         "serialized_period_us": period_us,
                         "by_number_of_rotating_sets": by_sets,
                         "frac_at_6_sets": (by_sets or {}).get("6", {}).get("frac") if isinstance(by_sets, dict) else None,
                         "kernel_us_at_6_sets": (by_sets or {}).get("6", {}).get("kernel_us") if isinstance(by_sets, dict) else None,
                         "by_number_of_rotating_sets_note": "the same lone launch rotating over the first k of up to 16 buffer sets of 100 MB (inputs AND outputs); inputs_vs_outputs "
                                                            "rotates them separately: the time follows the number of OUTPUT sets -- up to 6 x 40 MB the 256 MiB Infinity Cache absorbs "
                                                            "the stores and HBM carries the reads alone, from 12 sets on every byte is HBM's (frac_all_in_hbm); one set is what the chip's "
                                                            "fabric moves at best (fabric_ceiling_frac).  frac above is at --sets (default 8, as in every round); SURVEY 8(d) asks for >= 6",
                         "frac_all_in_hbm": (by_sets or {}).get("16", {}).get("frac") if isinstance(by_sets, dict) else None,
                         "fabric_ceiling_frac": (by_sets or {}).get("1", {}).get("frac") if isinstance(by_sets, dict) else None,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "overlapped": {"avg_launch_us": launch_us, "achieved": BYTES_PER_VERTEX * per_rank_verts / (launch_us * 1e-6) / 1e9,
                                        "frac": BYTES_PER_VERTEX * per_rank_verts / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                        "vertices_per_launch": per_rank_verts,
                                        "launch_streams": opts["lbs.streams"],
                                        "note": "time per launch of the timed region: consecutive launches overlap on the launch streams"},
                         "copy_ceiling": None if copy_us is None else {
                             "kernel_us": copy_us, "achieved": bytes_launch / (copy_us * 1e-6) / 1e9,
                             "frac": bytes_launch / (copy_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                             "note": "stream_copy_kernel: 60 MB read + 40 MB written, no math, one stream, same run"},
                         "position_only": pos_only},
            "parity": parity,
            "box": box,
        }
        extra = {}
        if random_rec is not None:
            extra["c4_random_bones"] = random_rec
        if strong is not None:
            extra["strong_scaling"] = strong
        if world == 1 and not args.no_extras and args.scaling == "weak":
            import subprocess
            sub = None
            try:
                ctx.sync()
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--extras-only"] + [f"--opt={kv}" for kv in args.opt],
                                    capture_output=True, text=True, timeout=900)
                lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                if cp.returncode == 0 and lines:
                    sub = json.loads(lines[-1])
                else:
                    print(f"# sub-records in a child process failed (rc {cp.returncode}): {cp.stderr[-400:]}", file=sys.stderr)
            except Exception as e:     # noqa: BLE001
                print(f"# sub-records in a child process failed: {e!r}", file=sys.stderr)
            extra.update(sub if sub is not None else extras(ctx))
            extra["sub_records_process"] = "child process (fresh HIP runtime)" if sub is not None else "this process"
        if extra:
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mesh, pal, args.cpu_seconds)
    else:
        out = None

    # ---- the crowd (BASELINE config 3) cut by instance range over the ranks: per-rank pose + skinning, no communication at all ----
    if (world > 1 or force_exchange) and not args.no_extras:
        i0, i1 = sharding.instance_range(1000, rank, world)
        rec, my_ms, crowd_err = None, float("inf"), None
        saved = {k: ctx.get_option(k) for k in ("lbs.streams", "anim.overlap", "lbs.crowd_lean")}
        try:
            ctx.set_option("lbs.streams", 1)
            rec = _c3_record(ctx, n_instances=i1 - i0, inst_offset=i0, frames=150, parity_instances=(0, i1 - i0 - 1), fused=False)
            my_ms = rec["frame_ms"]
        except BaseException as e:     # noqa: BLE001  (a parity failure raises SystemExit: reported, never fatal for the line)
            crowd_err = repr(e)
        for k, v in saved.items():
            ctx.set_option(k, v)
        f_ms = max_over_ranks(my_ms)          # every rank takes part, whatever happened to its own record
        if f_ms == float("inf"):
            crowd = {"error": crowd_err or "another rank's record failed"}
        else:
            crowd = {"workload": f"C3 cut by instance range over {world} GPU(s): every rank runs pose -> palettes -> skinning for its own "
                                 f"{i1 - i0} (rank 0) of 1000 instances x 10k verts / 64 bones; nothing is exchanged",
                     "scaling": "strong", "value": 1000 * 10_000 / (f_ms * 1e-3), "unit": "skinned vertices/s (whole frames, slowest rank)",
                     "frame_ms_slowest_rank": f_ms, "frame_ms_rank0": my_ms, "rank0_instances": [i0, i1], "rank0_record": rec}
        if rank == 0:
            out.setdefault("extra", {})["crowd_scaling"] = crowd
            if crowd and "value" in crowd:
                out["crowd_value"], out["crowd_frame_ms"] = crowd["value"], crowd["frame_ms_slowest_rank"]

    if rank == 0 and strong is not None and (world > 1 or force_exchange):
        # Both scalings at the top level whichever one `value` is: strong_value = BASELINE config 4 (the ONE 1 M-vertex mesh cut by vertex
        # range, compute only), weak_value = every GPU its own 1 M-vertex mesh, value_with_gather = config 4 with the RCCL exchange, the
        # fastest of the three exchange forms (all in extra.strong_scaling); crowd_value = config 3 cut by instance range, whole frames
        out["strong_value"] = strong["compute_only"]["value"]
        out["strong_ms_per_step"] = strong["compute_only"]["ms_per_step"]
        if weak_regions is not None:
            r_w, w_w, _ = weak_regions
            out["weak_value"] = float(world) * nv * args.steps * r_w / float(np.median(w_w))
            out["weak_ms_per_step"] = float(np.median(w_w)) * 1e3 / (args.steps * r_w)
        out["value_with_gather"], out["value_with_gather_form"] = None, None

    # ---- the exchange, last: RCCL with more than one rank has never run before the driver's multi-GPU job, so a hang in it
    # must not cost the line -- after EXCHANGE_TIMEOUT_S rank 0 prints what it has and every rank leaves.  All three forms are timed:
    # the driver's run is the A/B --------------------------------------------------------------------------------------------
    if strong is not None and have_comm and (world > 1 or force_exchange):
        import threading
        for form, key in ((0, "with_allgather"), (1, "with_allgather_sendrecv"), (2, "with_allgather_padded")):
            def give_up(key=key):
                if rank == 0:
                    out["extra"]["strong_scaling"][key] = {"value": None, "note": f"the exchange did not finish within {EXCHANGE_TIMEOUT_S} s"}
                    finish(out, args.full_record)
                os._exit(0)

            watchdog = threading.Timer(EXCHANGE_TIMEOUT_S, give_up)
            watchdog.daemon = True
            watchdog.start()
            try:
                rec, ok, regions = exchange_leg(form)
                err = None
            except Exception as e:     # noqa: BLE001
                rec, ok, regions, err = None, None, None, repr(e)
            watchdog.cancel()
            if rank == 0:
                st = out["extra"]["strong_scaling"]
                if rec is not None:
                    st[key] = rec
                    st["gathered_equals_oracle"] = ok if st.get("gathered_equals_oracle") in (None, True) else False
                    if ok is False:
                        st[key]["note"] = "THE GATHERED BUFFER DIFFERS FROM THE ORACLE"
                    elif out.get("value_with_gather") is None or rec["value"] > out["value_with_gather"]:
                        out["value_with_gather"], out["value_with_gather_form"] = rec["value"], ("broadcasts", "send_recv", "all_gather_padded")[form]
                    if args.scaling == "strong" and args.allgather and form == 0:
                        r_g, w_g, _ = regions
                        out["value"] = float(args.verts) * args.steps * r_g / float(np.median(w_g))
                        out["ms_per_step"] = float(np.median(w_g)) * 1e3 / (args.steps * r_g)
                        out["timed_steps"], out["repeats"], out["region_ms"] = args.steps * r_g, r_g, [w * 1e3 for w in w_g]
                else:
                    st[key] = {"value": None, "note": f"the exchange failed: {err}"}
            if err is not None and world > 1:
                break          # a failed collective leaves the communicator in an unknown state: no second form
        ctx.set_option("comm.form", 0)
    if rank == 0:
        if isinstance(out.get("box"), dict):
            out["box"]["at_end"] = box_facts(local_rank)
            out["box"]["note"] = ("sysfs of the amdgpu card + hwmon + rocm-smi at the start of the run, right behind the headline's timed regions and at the "
                                  "end: clocks (the level marked * is the current one), power now / cap (microwatts), temperatures (millidegrees), "
                                  "compute / memory partition modes, driver versions -- what tells a slow box from a fast one")
        finish(out, args.full_record)

    barrier()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
