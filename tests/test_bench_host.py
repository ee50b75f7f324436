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
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.scaling == "weak"
    argv, sys.argv = sys.argv, ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    try:
        a = b.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
