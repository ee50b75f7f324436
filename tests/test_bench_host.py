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
