"""The parts of bench.py that need no GPU (the driver runs bench.py on an MI355X; these keep its host-only pieces
from rotting between GPU runs)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("fyx_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_host_control_plane_record_runs_without_a_gpu():
    rec = _bench()._host_control_plane_record(frames=5)
    assert "error" not in rec, rec
    assert 0.0 < rec["c3_plan_us_with_memo"] < rec["c3_plan_us_from_scratch"] * 1.5
    assert rec["scene_256x1_plan_us"] > 0.0


def test_argument_parser_defaults_match_the_driver_contract():
    b = _bench()
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = b.parse()
    finally:
        sys.argv = argv
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.scaling is None      # None: weak at N = 1, strong (BASELINE config 4) at N > 1
    argv, sys.argv = sys.argv, ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    try:
        a = b.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)


def test_a_plain_launch_with_several_gpus_becomes_the_drivers_launch_line():
    """`python bench.py --gpus 8` without a launcher must not measure one GPU under the label of eight: it re-executes
    itself under torch.distributed.run with one rank per GPU, and a world size that does not match --gpus is refused."""
    import pytest
    b = _bench()
    cmd = b.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    b.check_world(8, 8)
    b.check_world(1, 1)
    for n, w in ((8, 1), (1, 2), (4, 8)):
        with pytest.raises(SystemExit):
            b.check_world(n, w)


def test_main_refuses_a_mismatching_world_before_it_touches_a_gpu():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert cp.returncode != 0 and "WORLD_SIZE=1" in (cp.stderr + cp.stdout) and "n_gpus" not in cp.stdout


def test_the_launcher_road_reports_failure_so_that_the_one_process_road_is_taken():
    """The plain `--gpus N` launch runs torch.distributed.run as a CHILD under a timeout; a child that fails, prints no line, cannot be
    started or does not finish makes run_under_launcher return no line and a reason -- main() then runs main_one_process (all N GPUs
    from this process through fyx_comm_init_all / fyx_allgather_skinned_all) and says so in config.process_group."""
    import subprocess
    b = _bench()
    good = '{"metric": "skinned vertices/sec", "value": 1.0, "n_gpus": 2}'

    class CP:
        def __init__(self, rc, out, err=""):
            self.returncode, self.stdout, self.stderr = rc, out, err

    line, why = b.run_under_launcher(["x"], 5.0, runner=lambda *a, **k: CP(0, "# noise\n" + good + "\n"))
    assert line == good and why == "ok"
    line, why = b.run_under_launcher(["x"], 5.0, runner=lambda *a, **k: CP(1, good))
    assert line is None and "exited with 1" in why
    line, why = b.run_under_launcher(["x"], 5.0, runner=lambda *a, **k: CP(0, "nothing json here"))
    assert line is None and "no line" in why

    def hangs(*a, **k):
        raise subprocess.TimeoutExpired(cmd="x", timeout=5.0)
    line, why = b.run_under_launcher(["x"], 5.0, runner=hangs)
    assert line is None and "did not finish" in why

    def missing(*a, **k):
        raise FileNotFoundError("no such interpreter")
    line, why = b.run_under_launcher(["x"], 5.0, runner=missing)
    assert line is None and "could not be started" in why
    # a real child that fails (the real subprocess.run)
    line, why = b.run_under_launcher([sys.executable, "-c", "import sys; sys.exit(3)"], 60.0)
    assert line is None and "exited with 3" in why


def test_one_process_flag_is_parsed():
    b = _bench()
    argv, sys.argv = sys.argv, ["bench.py", "--gpus", "2", "--one-process", "--launcher-timeout", "7"]
    try:
        a = b.parse()
    finally:
        sys.argv = argv
    assert a.one_process and a.gpus == 2 and a.launcher_timeout == 7.0


def test_the_line_is_a_digest_the_driver_can_parse():
    """round 5's line had grown to 34 KB and the driver's parser gave up (`parsed: null`).  The line is now a digest of
    the full record -- the contract's keys, roofline with the measured traffic, parity, cpu_baseline, numbers of the sub-records --
    and stays under LINE_LIMIT whatever the full record holds; the full record goes to a file the line names."""
    import json
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_plain_final.json")))       # round 5's 34 KB record, as the driver saw it
    assert len(json.dumps(full)) > 30_000
    line = b.compact_line(full, "bench_full.json")
    assert len(line) < b.LINE_LIMIT <= 8192 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["metric"] == full["metric"] and d["n_gpus"] == 1 and d["dtype"] == "f32" and "C4" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] / full["value"] - 1.0) < 1e-5 and abs(d["ms_per_step"] / full["ms_per_step"] - 1.0) < 1e-4
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "lbs_skin_dyn"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and abs(r["frac"] / full["roofline"]["frac"] - 1.0) < 1e-3
    assert r["traffic"] > 1e8 and r["traffic_source"] == "pmc_in_this_run" and r["algorithmic_bytes_per_launch"] == 100_000_000
    assert r["frac_at_6_sets"] > 0.7 and r["overlapped"]["frac"] > r["frac"] and 0.6 < r["copy_ceiling"]["frac"] < 0.8
    assert d["parity"] == {"max_rel_err": 0.0, "bit_exact": True, "checked_vertices": 1_000_000}
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 1e6 and cb["unit"] == "vertices/s" and cb["sample"]
    g = d["digest"]
    for k in ("c2_frame_ms", "c5_frame_ms", "c3_frame_ms", "c3_crowd_kernel_frac", "scene_256x1_frame_ms", "scene_64x4_frame_ms", "vb_plain_frac"):
        assert isinstance(g[k], float), k
    assert all(isinstance(v, (int, float, bool)) for v in g.values())        # numbers only, no prose
    assert d["full_record"] == "bench_full.json"
    # whatever the full record grows into, the line stays short: the digest is dropped before the contract's keys are
    fat = dict(full, extra=dict(full["extra"], **{f"c{i}": full["extra"]["c3"] for i in range(6, 400)}))
    assert len(b.compact_line(fat, None)) < b.LINE_LIMIT
    orig = b.LINE_LIMIT
    try:
        b.LINE_LIMIT = 2200
        short = json.loads(b.compact_line(full, None))
        assert "digest" not in short and "digest" in short["dropped_for_length"] and short["roofline"]["frac"] == r["frac"] and "cpu_baseline" in short
    finally:
        b.LINE_LIMIT = orig


def test_the_line_of_several_gpus_carries_both_scalings():
    """N > 1: `value` is BASELINE config 4 as written (strong scaling, compute only); weak_value, value_with_gather and
    the crowd's number ride beside it, and the exchange legs keep their numbers."""
    import json
    b = _bench()
    full = {"metric": "m", "value": 2.0e11, "unit": "vertices/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 0.005, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "timed_steps": 8000, "repeats": 400,
            "config": {"workload": "C4 as written ...", "sharding": "contiguous vertex range per GPU, palette replicated", "n_ranks": 8, "process_group": "nccl", "sets": 8,
                       "kernel_options": {"lbs.dyn": 1}},
            "roofline": {"bound": "hbm", "achieved": 5300.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.6625, "traffic": None, "traffic_source": None, "kernel": "lbs_skin_dyn", "kernel_us": 18.9},
            "parity": {"max_rel_err": 0.0, "bit_exact": True, "checked_vertices": 1000000, "streams_checked": ["pos"]},
            "strong_value": 2.0e11, "strong_ms_per_step": 0.005, "weak_value": 4.9e11, "weak_ms_per_step": 0.0163, "value_with_gather": 3.1e9, "value_with_gather_form": "send_recv",
            "crowd_value": 7.7e11, "crowd_frame_ms": 0.013,
            "extra": {"strong_scaling": {"compute_only": {"value": 2.0e11, "ms_per_step": 0.005}, "with_allgather": {"value": 2.2e9, "ms_per_step": 0.45, "collective": "x" * 300},
                                         "with_allgather_sendrecv": {"value": 3.1e9, "ms_per_step": 0.32}, "with_allgather_padded": {"value": None, "note": "timed out"},
                                         "gathered_equals_oracle": True, "shard_vertices": [125000] * 8}}}
    d = json.loads(b.compact_line(full, "bench_full.json"))
    assert d["scaling"] == "strong" and d["value"] == d["strong_value"] == 2.0e11 and d["weak_value"] == 4.9e11 and d["value_with_gather_form"] == "send_recv"
    assert d["exchange_legs"]["with_allgather_sendrecv"] == {"value": 3.1e9, "ms_per_step": 0.32} and d["exchange_legs"]["with_allgather_padded"]["value"] is None
    assert "compute_only" not in d["exchange_legs"] and d["gathered_equals_oracle"] is True and d["crowd_value"] == 7.7e11
    assert "kernel_options" not in d["config"] and d["config"]["n_ranks"] == 8
