"""In-tree build of ``pytorch_distributed_training_amd/_C*.so`` for gfx950.

Plain ``hipcc`` (device code, ``--offload-arch=gfx950``) + host C++ against the
installed libtorch, no hipify pass, no JIT cache: the built ``.so`` sits next
to the package so it travels with the tree to the GPU box.

    python -m pytorch_distributed_training_amd.csrc.build [--force] [-j N]
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys
import sysconfig
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
BUILD = os.path.join(HERE, "_build")
ARCH = "gfx950"

HIP_SOURCES = [
    "kernels/elementwise.hip", "kernels/softmax_ce.hip", "kernels/bn.hip", "kernels/layernorm.hip", "kernels/attention.hip", "kernels/pool.hip",
    "kernels/gemm.hip", "kernels/gemm_tn.hip", "kernels/conv_win64.hip",
    "kernels/conv_stem_win.hip", "kernels/wgrad_win64.hip", "kernels/vit.hip", "kernels/conv_igemm.hip", "kernels/conv_wgrad.hip", "kernels/conv_wgrad_win.hip", "kernels/conv_stem.hip",
    "comm/ipc_allreduce.hip",
]
# Kernels whose job is numerics (statistics, loss, optimizer, pooling, normalisation) keep NaN / Inf semantics:
# -ffast-math implies -ffinite-math-only, under which the compiler may fold away the very comparisons and clamps
# that decide whether a NaN stays visible.  They keep the rest of fast-math (contraction,
# reassociation, approximate reciprocals); the GEMM-shaped kernels keep plain -ffast-math.
NAN_AWARE = {"kernels/bn.hip", "kernels/softmax_ce.hip", "kernels/elementwise.hip", "kernels/pool.hip",
             "kernels/layernorm.hip", "kernels/vit.hip", "comm/ipc_allreduce.hip"}
CPP_SOURCES = ["binding/ops.cpp", "binding/module.cpp", "comm/rccl_comm.cpp", "comm/ipc_allreduce.cpp", "reducer/reducer.cpp"]


def _torch_paths():
    import torch
    from torch.utils import cpp_extension as ce

    inc = ce.include_paths("cuda")
    lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    return inc, lib


def ext_path() -> str:
    return os.path.join(PKG, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def _headers_digest() -> str:
    h = hashlib.sha1()
    for root, _, files in os.walk(HERE):
        if "_build" in root:
            continue
        for f in sorted(files):
            if f.endswith(".h"):
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()[:12]


def _needs(src: str, obj: str, stamp: str) -> bool:
    if not os.path.exists(obj) or not os.path.exists(obj + ".stamp"):
        return True
    if open(obj + ".stamp").read() != stamp:
        return True
    return os.path.getmtime(src) > os.path.getmtime(obj)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed:\n  " + " ".join(cmd) + "\n" + r.stdout)
    if r.stdout.strip():
        sys.stderr.write(r.stdout)


def build(force: bool = False, jobs: int | None = None, verbose: bool = True) -> str:
    os.makedirs(BUILD, exist_ok=True)
    inc, torch_lib = _torch_paths()
    py_inc = sysconfig.get_paths()["include"]
    digest = _headers_digest()
    jobs = jobs or min(8, os.cpu_count() or 4)
    common_defs = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2", "-DTORCH_EXTENSION_NAME=_C",
                   "-D_GLIBCXX_USE_CXX11_ABI=1", "-DTORCH_API_INCLUDE_EXTENSION_H"]
    hip_flags = ["hipcc", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-gpu-rdc",
                 "-Wno-unused-result"]
    cpp_flags = ["hipcc", "-O2", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-Wno-deprecated-declarations",
                 *common_defs, *[f"-I{p}" for p in inc], f"-I{py_inc}"]
    tasks = []
    objs = []
    for rel in HIP_SOURCES + CPP_SOURCES:
        src = os.path.join(HERE, rel)
        if not os.path.exists(src):
            continue
        obj = os.path.join(BUILD, rel.replace("/", "_") + ".o")
        objs.append(obj)
        is_hip = rel.endswith(".hip")
        flags = hip_flags if is_hip else cpp_flags
        if rel in NAN_AWARE:
            flags = [*flags, "-fno-finite-math-only"]
        stamp = digest + "|" + " ".join(flags)
        if force or _needs(src, obj, stamp):
            cmd = [*flags, "-x", "hip" if is_hip else "c++", "-c", src, "-o", obj]  # .cpp files are host-only
            tasks.append((cmd, obj, stamp, rel))
    t0 = time.time()
    if tasks:
        def work(t):
            cmd, obj, stamp, rel = t
            t1 = time.time()
            _run(cmd)
            with open(obj + ".stamp", "w") as fh:
                fh.write(stamp)
            if verbose:
                print(f"[build] {rel} ({time.time() - t1:.1f}s)", flush=True)

        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(work, tasks))
    out = ext_path()
    if tasks or not os.path.exists(out) or force:
        link = ["hipcc", "-shared", "-fPIC", "-o", out, *objs, f"-L{torch_lib}", "-L/opt/rocm/lib",
                "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python", "-lrccl",
                "-lamdhip64", f"-Wl,-rpath,{torch_lib}", "-Wl,-rpath,/opt/rocm/lib"]
        _run(link)
        if verbose:
            print(f"[build] linked {out} ({time.time() - t0:.1f}s total)", flush=True)
    elif verbose:
        print(f"[build] up to date: {out}", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=None)
    a = ap.parse_args()
    build(a.force, a.j)


if __name__ == "__main__":
    main()
