"""A compiled C++ host (tests/cpp/host_parity.cpp) drives the library through the C ABI alone -- as the engine's Rust
side would -- and checks it against the oracle linked into the test binary."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_parity")


def _build():
    import oracle
    oracle.lib()   # makes sure oracle/libfyrox_oracle.so exists
    src = os.path.join(ROOT, "tests", "cpp", "host_parity.cpp")
    libs = [os.path.join(ROOT, "fyrox_amd", "libfyrox_hip.so"), os.path.join(ROOT, "oracle", "libfyrox_oracle.so")]
    if not os.path.exists(BIN) or any(os.path.getmtime(BIN) < os.path.getmtime(p) for p in [src] + libs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", src, "-o", BIN,
                               f"-L{ROOT}/fyrox_amd", f"-L{ROOT}/oracle", "-lfyrox_hip", "-lfyrox_oracle",
                               f"-Wl,-rpath,{ROOT}/fyrox_amd", f"-Wl,-rpath,{ROOT}/oracle", "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


def test_cpp_host_control_plane_matches_oracle():
    out = subprocess.run([_build(), "--control-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "control plane ok" in out.stdout


@pytest.mark.gpu
def test_cpp_host_c1_and_clip_parity_on_gpu():
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu ok" in out.stdout
