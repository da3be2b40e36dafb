"""Profiler wrapper: torch.profiler schedule of the reference + roctx ranges.

Reference (``/root/reference/main.py:68-78,115``): ``torch.profiler.profile(
schedule(wait=2, warmup=2, active=6, repeat=1), tensorboard_trace_handler(
f'./log_{JobID}'), activities=[CPU, CUDA], with_stack=True)`` around the whole
run, ``p.step()`` per iteration: steps 0-1 NONE, 2-3 WARMUP, 4-8 RECORD, 9
RECORD_AND_SAVE, then NONE.

``--profile torch`` reproduces that exactly (``ProfilerActivity.CUDA`` is the
HIP device on ROCm, traced through roctracer).  ``--profile roctx`` emits
``roctxRangePush/Pop`` markers (libroctx64) around forward / backward /
optimizer so rocprofv3 timelines are readable; ``--profile none`` is free.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch

SCHEDULE = dict(wait=2, warmup=2, active=6, repeat=1)


def schedule_action(step: int, wait=2, warmup=2, active=6, repeat=1) -> str:
    """Pure-Python model of the schedule (unit-tested against torch.profiler.schedule)."""
    cycle = wait + warmup + active
    if repeat > 0 and step >= repeat * cycle:
        return "NONE"
    m = step % cycle
    if m < wait:
        return "NONE"
    if m < wait + warmup:
        return "WARMUP"
    return "RECORD" if m < cycle - 1 else "RECORD_AND_SAVE"


class _Roctx:
    def __init__(self):
        self.lib = None
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so"):
            try:
                self.lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if self.lib is not None:
            self.lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            self.lib.roctxRangePushA.restype = ctypes.c_int
            self.lib.roctxRangePop.restype = ctypes.c_int
            self.lib.roctxMarkA.argtypes = [ctypes.c_char_p]

    def push(self, msg: str) -> None:
        if self.lib is not None:
            self.lib.roctxRangePushA(msg.encode())

    def pop(self) -> None:
        if self.lib is not None:
            self.lib.roctxRangePop()


_ROCTX: _Roctx | None = None
_ROCTX_ENABLED = False


def enable_roctx(flag: bool = True) -> bool:
    global _ROCTX, _ROCTX_ENABLED
    if flag and _ROCTX is None:
        _ROCTX = _Roctx()
    _ROCTX_ENABLED = bool(flag) and _ROCTX is not None and _ROCTX.lib is not None
    return _ROCTX_ENABLED


def roctx_enabled() -> bool:
    return _ROCTX_ENABLED


@contextlib.contextmanager
def range_(name: str):
    """roctx range when enabled, otherwise free."""
    if _ROCTX_ENABLED:
        _ROCTX.push(name)
        try:
            yield
        finally:
            _ROCTX.pop()
    else:
        yield


class NullProfiler:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def step(self):
        pass


def make_profiler(mode: str, job_id: str, with_stack: bool = True):
    """``mode`` in {none, torch, roctx}."""
    mode = (mode or "none").lower()
    if mode == "none":
        return NullProfiler()
    if mode == "roctx":
        enable_roctx(True)
        return NullProfiler()
    if mode != "torch":
        raise ValueError(f"unknown --profile mode {mode!r}")
    import torch.profiler as tp

    acts = [tp.ProfilerActivity.CPU]
    if torch.cuda.is_available():
        acts.append(tp.ProfilerActivity.CUDA)
    return tp.profile(
        schedule=tp.schedule(**SCHEDULE),
        on_trace_ready=tp.tensorboard_trace_handler(dir_name=os.path.join(".", f"log_{job_id}")),
        activities=acts,
        with_stack=with_stack,
    )
