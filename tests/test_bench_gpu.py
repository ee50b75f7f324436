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


@pytest.mark.gpu
def test_plain_launch_with_two_ranks_prints_every_multi_gpu_key():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FYX_BENCH_DEVICE"] = "0"
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--sets", "3"],
                        env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert cp.returncode == 0 and lines, (cp.returncode, cp.stdout[-2000:], cp.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "WEAK scaling" in d["config"]["workload"] and "strong_value" in d["config"]["workload"]
    for key in ("strong_value", "strong_ms_per_step", "strong_with_gather_value", "strong_with_gather_form", "crowd_value", "crowd_frame_ms", "top_level_note"):
        assert key in d, key
    st = d["extra"]["strong_scaling"]
    assert st["compute_only"]["value"] > 0 and sum(st["shard_vertices"]) == 1_000_000
    # the exchange legs: numbers when RCCL accepted the communicator, otherwise the reason -- never silence
    if d["strong_with_gather_value"] is None:
        assert st.get("comm_error") or all(st[k].get("note") for k in ("with_allgather",))
    else:
        for k in ("with_allgather", "with_allgather_sendrecv", "with_allgather_padded"):
            assert k in st
    assert d["extra"]["crowd_scaling"].get("value", 0) > 0 or "error" in d["extra"]["crowd_scaling"]


@pytest.mark.gpu
def test_one_process_road_prints_the_line_and_says_why_there_is_no_exchange():
    """`python bench.py --gpus 2 --one-process` on a one-GPU box (FYX_BENCH_DEVICE=0: both contexts on GPU 0): no launcher, no torch
    process group -- the line carries n_gpus = 2, the weak and strong compute legs, and the refusal fyx_comm_init_all gives two
    contexts on one device in comm_error (the exchange legs need two GPUs)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FYX_BENCH_DEVICE"] = "0"
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-process", "--steps", "20", "--warmup", "5", "--sets", "2"],
                        env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert cp.returncode == 0 and lines, (cp.returncode, cp.stdout[-2000:], cp.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["strong_value"] > 0
    assert "ONE process" in d["config"]["process_group"] and "--one-process" in d["config"]["process_group"]
    assert d["parity"]["bit_exact"] is True
    assert d["strong_with_gather_value"] is None and "same GPU" in d["comm_error"]
    assert "at_start" in d["box"] and "at_end" in d["box"]


@pytest.mark.gpu
def test_the_headline_line_carries_the_box_facts():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline", "--sets", "3"],
                        env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert cp.returncode == 0 and lines, (cp.returncode, cp.stdout[-2000:], cp.stderr[-3000:])
    d = json.loads(lines[-1])
    for when in ("at_start", "after_headline", "at_end"):
        assert when in d["box"], when
    assert d["box"]["hip_device"].get("name")
    # roofline.traffic: measured in this run by two rocprofv3 --pmc passes (or, when rocprofv3 cannot run here, the reason and the labelled replay)
    tm = d["roofline"]["traffic_measurement"]
    assert isinstance(tm, dict)
    if "error" not in tm:
        assert 0.95e8 < tm["hbm_bytes_per_launch"] < 1.2e8, tm
        assert d["roofline"]["traffic"] == tm["hbm_bytes_per_launch"] and "measured in this run" in d["roofline"]["traffic_source"]
    else:
        assert d["roofline"]["traffic"] is None or "not measured in this run" in d["roofline"]["traffic_source"]
