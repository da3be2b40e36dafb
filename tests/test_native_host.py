"""SURVEY 5.2 (race detection / sanitizers): the target pool has no GPU sanitizers, so the host-side C++ that the
native runtime shares with the kernels' launch math (bucket planner, fast-division magic numbers) is built with
AddressSanitizer + UBSan and run here on the CPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pytorch_distributed_training_amd", "csrc", "tests", "host_sanitize_test.cpp")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_code_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize_test"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", SRC, "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)  # ASan must be the first runtime in the process it instruments
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "OK" in r.stdout
