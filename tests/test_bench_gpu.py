"""bench.py's N > 1 code on a one-GPU box: `python bench.py --gpus 2` launched PLAINLY (no torchrun) with the test hook
FYX_BENCH_DEVICE=0 (every rank on GPU 0).  The first real multi-GPU run is the driver's and cannot be rehearsed; what can be
checked here is that a plain launch becomes two ranks, that the line says n_gpus = 2 and that every multi-GPU key is there (the
RCCL legs may report the refusal RCCL gives two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env, tmp_path):
    """bench.py's last stdout line (the compact record the driver parses) and the full record it names."""
    full = str(tmp_path / "bench_full.json")
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--full-record", full], env=env, capture_output=True, text=True, timeout=900)
    out_lines = cp.stdout.splitlines()
    assert cp.returncode == 0 and out_lines and out_lines[-1].startswith("{"), (cp.returncode, cp.stdout[-2000:], cp.stderr[-3000:])
    assert len(out_lines[-1]) < 6144, len(out_lines[-1])
    d = json.loads(out_lines[-1])
    assert d["full_record"]
    return d, json.load(open(full))


@pytest.mark.gpu
def test_plain_launch_with_two_ranks_prints_every_multi_gpu_key(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FYX_BENCH_DEVICE"] = "0"
    d, full = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--sets", "3"], env, tmp_path)
    # N > 1: `value` is BASELINE config 4 as written -- ONE 1 M-vertex mesh cut by vertex range, compute only
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["value"] == pytest.approx(d["strong_value"], rel=1e-4) and d["weak_value"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / 1_000_000 - 1.0) < 1e-3          # one step = the whole 1 M-vertex mesh
    assert "STRONG-scaling" in d["config"]["workload"] and "value_with_gather" in d["config"]["workload"]
    for key in ("strong_value", "strong_ms_per_step", "weak_value", "weak_ms_per_step", "value_with_gather", "value_with_gather_form", "crowd_value", "crowd_frame_ms"):
        assert key in d, key
        assert key in full, key
    st = full["extra"]["strong_scaling"]
    assert st["compute_only"]["value"] > 0 and sum(st["shard_vertices"]) == 1_000_000
    # the exchange legs: numbers when RCCL accepted the communicator, otherwise the reason -- never silence
    if d["value_with_gather"] is None:
        assert st.get("comm_error") or all(st[k].get("note") for k in ("with_allgather",))
    else:
        for k in ("with_allgather", "with_allgather_sendrecv", "with_allgather_padded"):
            assert k in st
    assert full["extra"]["crowd_scaling"].get("value", 0) > 0 or "error" in full["extra"]["crowd_scaling"]


@pytest.mark.gpu
def test_one_process_road_prints_the_line_and_says_why_there_is_no_exchange(tmp_path):
    """`python bench.py --gpus 2 --one-process` on a one-GPU box (FYX_BENCH_DEVICE=0: both contexts on GPU 0): no launcher, no torch
    process group -- the line carries n_gpus = 2, the weak and strong compute legs, and the refusal fyx_comm_init_all gives two
    contexts on one device in comm_error (the exchange legs need two GPUs)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FYX_BENCH_DEVICE"] = "0"
    d, full = _run(["--gpus", "2", "--one-process", "--steps", "20", "--warmup", "5", "--sets", "2"], env, tmp_path)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["value"] == pytest.approx(d["strong_value"], rel=1e-4) and d["weak_value"] > 0
    assert "ONE process" in d["config"]["process_group"] and "--one-process" in d["config"]["process_group"]
    assert d["parity"]["bit_exact"] is True
    assert d["value_with_gather"] is None and "same GPU" in d["comm_error"]
    assert "at_start" in full["box"] and "at_end" in full["box"]


@pytest.mark.gpu
def test_the_headline_line_is_short_and_its_full_record_carries_the_box_facts(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    line, d = _run(["--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline", "--sets", "3"], env, tmp_path)
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["dtype"] == "f32"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_us"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-3) and line["parity"]["bit_exact"] is True
    assert line["box"]["cus"] == 256
    for when in ("at_start", "after_headline", "at_end"):
        assert when in d["box"], when
    assert d["box"]["hip_device"].get("name")
    # roofline.traffic: measured in this run by two rocprofv3 --pmc passes (or, when rocprofv3 cannot run here, the reason and the labelled replay)
    tm = d["roofline"]["traffic_measurement"]
    assert isinstance(tm, dict)
    if "error" not in tm:
        assert 0.95e8 < tm["hbm_bytes_per_launch"] < 1.2e8, tm
        assert d["roofline"]["traffic"] == tm["hbm_bytes_per_launch"] and "measured in this run" in d["roofline"]["traffic_source"]
        assert line["roofline"]["traffic_source"] == "pmc_in_this_run" and line["roofline"]["traffic"] == pytest.approx(tm["hbm_bytes_per_launch"], rel=1e-5)
    else:
        assert d["roofline"]["traffic"] is None or "not measured in this run" in d["roofline"]["traffic_source"]
