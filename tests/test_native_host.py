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


# ---------------------------------------------------------------------------------------------------------------
# The C++ reducer + communicator host logic under ASan/UBSan and under TSan (csrc/tests/reducer_sanitize_test.cpp):
# rank threads in one instrumented process, an in-process c10d group, a second thread polling the reducer's API while
# the autograd engine drives its hooks.
CSRC = os.path.join(ROOT, "pytorch_distributed_training_amd", "csrc")


def _sanitizer_build(tmp_path, name, san_flags):
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce

    inc = ce.include_paths("cuda")
    lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    exe = str(tmp_path / name)
    cmd = ["hipcc", "-x", "c++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-w", *san_flags,
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=1", *[f"-I{p}" for p in inc],
           f"-I{sysconfig.get_paths()['include']}",
           os.path.join(CSRC, "tests", "reducer_sanitize_test.cpp"), os.path.join(CSRC, "reducer", "reducer.cpp"),
           os.path.join(CSRC, "comm", "rccl_comm.cpp"), os.path.join(CSRC, "comm", "ipc_allreduce.cpp"), "-o", exe, f"-L{lib}", "-ltorch", "-ltorch_cpu",
           "-ltorch_hip", "-lc10", "-lc10_hip", "-lrccl", "-lamdhip64", "-lpthread", "-ldl", f"-Wl,-rpath,{lib}"]  # (torch's own ROCm libraries only: a second copy from /opt/rocm double-frees at exit)
    return exe, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_reducer_and_comm_host_logic_under_asan_ubsan_and_tsan(tmp_path):
    builds = {
        "asan": _sanitizer_build(tmp_path, "reducer_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
        "tsan": _sanitizer_build(tmp_path, "reducer_tsan", ["-fsanitize=thread"]),
    }
    for kind, (exe, proc) in builds.items():
        out, _ = proc.communicate(timeout=900)
        assert proc.returncode == 0, f"{kind} build failed:\n{out[-3000:]}"
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)  # a sanitizer runtime must be the first runtime in the process it instruments
    env["OMP_NUM_THREADS"] = "1"
    # libtorch itself is not instrumented: leaks inside it are not ours to report, and TSan must not report races
    # between two uninstrumented frames; everything that touches reducer.cpp / rccl_comm.cpp / the test IS reported
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["TSAN_OPTIONS"] = "halt_on_error=1:report_signal_unsafe=0:ignore_noninstrumented_modules=1"
    for kind, (exe, _) in builds.items():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "OK" in r.stdout, f"{kind} run:\n" + (r.stdout + r.stderr)[-4000:]
