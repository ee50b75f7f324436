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
