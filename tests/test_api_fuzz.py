"""A few pinned seeds of the three API-sequence fuzzers (tools/fuzz_api_host.py, fuzz_api_gpu.py, fuzz_api_pose_gpu.py) inside the suites: calls
with wrong arguments -- stale, freed and out-of-range ids, null pointers, palettes too short, members listed twice, frees of objects in use --
earn error codes and nothing else: the process survives, a refused data-path call writes nothing, and the next good call or frame is the
oracle's bit for bit.  The long runs are recorded in profiles/r06_fuzz/."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_api_sequences_on_a_device_less_context():
    import fuzz_api_host
    calls, errors = 0, {}
    for seed in range(12):
        r = fuzz_api_host.one_sequence(seed)
        calls += r["calls"]
        for k, v in r["errors"].items():
            errors[k] = errors.get(k, 0) + v
    assert calls > 400 and set(errors) <= fuzz_api_host.EARNED and sum(errors.values()) > 50, (calls, errors)


@pytest.mark.gpu
def test_api_sequences_on_the_data_path(ctx):
    import fuzz_api_gpu
    stats = {"calls": 0, "errors": {}, "refused": 0, "good_calls_checked": 0}
    for seed in range(4):
        fuzz_api_gpu.one_sequence(ctx, seed, stats)
    assert stats["refused"] > 20 and stats["good_calls_checked"] == stats["refused"] and set(stats["errors"]) <= fuzz_api_gpu.EARNED, stats


@pytest.mark.gpu
def test_api_sequences_on_the_pose_path(ctx):
    import fuzz_api_pose_gpu
    stats = {"calls": 0, "errors": {}, "refused": 0, "frames_checked": 0}
    for seed in range(25):
        fuzz_api_pose_gpu.one_sequence(ctx, seed, stats)
    assert stats["refused"] > 300 and stats["frames_checked"] > 200 and set(stats["errors"]) <= fuzz_api_pose_gpu.EARNED, stats
