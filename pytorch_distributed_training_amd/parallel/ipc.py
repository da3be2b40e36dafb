"""Latency-path all-reduce for SyncBatchNorm: one kernel over IPC-mapped peer memory.

SURVEY §2.5 S4a/S4b, §5.8 layer 3, §7.4 item 4: SyncBN issues 106 data-dependent
all-reduces of at most 2C+4 floats per step.  They cannot be overlapped, only made
cheap.  Through a process-group collective each one costs a launch on RCCL's stream
plus its proxy hand-shake; ``_C.IpcAllReduce`` (``csrc/comm/ipc_allreduce.*``) does
the same sum in ONE kernel on the compute stream: every rank stores its values
straight into the mailboxes of all peers over xGMI and adds what arrived in its own.

``small_all_reduce`` is what ``nn/functional.py`` calls.  Which path it takes is a
per-group decision made once, collectively, at first use (``PDT_SYNCBN_COMM``):

``auto`` (default)  set the mailboxes up, check the kernel's result against the
                    process group's on random data, time both, keep the faster one;
                    any failure on any rank (IPC refused, ranks on different hosts,
                    mismatch, time-out) sends every rank to the process group.
``ipc``             same checks, but IPC is used whenever it is correct.
``torch`` / ``rccl``  the process group's all-reduce (what ``torch.nn.SyncBatchNorm``
                    does, ``torch/nn/modules/_functions.py:48-77,157-165``).
"""
from __future__ import annotations

import os
import socket
import sys
import time

import torch
import torch.distributed as tdist

_CAPACITY = 16384  # floats per message (2C+4 at C=2048 is 4100)
_paths: dict = {}


def _ext():
    from ..utils.env import native

    return native()


class IpcSmallAllReduce:
    """The mailboxes of one process group.  Construction is collective: every rank of ``group`` must call it."""

    def __init__(self, group=None, device: torch.device | None = None, capacity: int = _CAPACITY,
                 timeout_s: float | None = None):
        self.group = group
        self.rank = tdist.get_rank(group)
        self.world = tdist.get_world_size(group)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.capacity = capacity
        if timeout_s is None:
            timeout_s = float(os.environ.get("PDT_IPC_TIMEOUT_S", "120"))
        self.timeout_s = timeout_s
        self.native = None
        err = None
        handle = None
        try:
            ext = _ext()
            if ext is None or not hasattr(ext, "IpcAllReduce"):
                raise RuntimeError("native extension not built")
            self.native = ext.IpcAllReduce(self.rank, self.world, self.device.index or 0, capacity, timeout_s)
            handle = bytes(self.native.handle())
        except Exception as e:  # every rank still takes part in the exchange below
            err = repr(e)
        mine = (handle, socket.gethostname(), err)
        everyone = [None] * self.world
        tdist.all_gather_object(everyone, mine, group=group)
        if err is None:
            try:
                bad = [f"rank {r}: {e[2]}" for r, e in enumerate(everyone) if e[2] is not None]
                if bad:
                    raise RuntimeError("; ".join(bad))
                if len({e[1] for e in everyone}) != 1:
                    raise RuntimeError("the group spans hosts: peer memory cannot be mapped")
                self.native.open_peers([e[0] for e in everyone])
            except Exception as e:
                err = repr(e)
        self.error = self._agree(err)
        if self.error is not None:
            self.native = None

    def _agree(self, err):
        """None when every rank is fine, else one rank's message on every rank."""
        msgs = [None] * self.world
        tdist.all_gather_object(msgs, err, group=self.group)
        bad = [f"rank {r}: {m}" for r, m in enumerate(msgs) if m is not None]
        return "; ".join(bad) if bad else None

    @property
    def ok(self) -> bool:
        return self.native is not None

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        self.native.all_reduce_(t)
        return t

    def timed_out(self) -> bool:
        return self.native is not None and self.native.timed_out()

    def close(self, collective: bool = True) -> None:
        """Two phases: device synchronised and the peers' mappings closed; a barrier over the group (an exporter
        must not free memory that a peer still has mapped); own mailbox freed.  Collective unless
        ``collective=False`` (a rank that tears down alone after its peers died: no barrier).  Call it only once
        every rank is past its last exchange (``dist.shutdown`` does, behind its own barrier): a peer's kernel
        that still stores into this mailbox would fault."""
        if self.native is None:
            return
        self.native.close_peers()
        if collective and self.world > 1:
            if tdist.get_backend(self.group) == "nccl":
                tdist.barrier(group=self.group, device_ids=[self.device.index or 0])
            else:
                tdist.barrier(group=self.group)
        self.native.close()
        self.native = None

    def _group_all_reduce(self, t: torch.Tensor) -> None:
        _all_reduce_group(t, self.group)  # (host-staged under gloo: tests drive GPU mailboxes from a CPU group)

    # -- collective self-check + timing --------------------------------------------------------------
    def validate(self, sizes=(132, 1028, 4100), iters: int = 20, check_timeout_s: float = 5.0):
        """-> (error or None, ipc_us, group_us): compares against the process group's all-reduce on random data
        and times both.  Collective: every rank runs the same sequence of process-group calls whatever happens
        to its IPC path, and a rank whose IPC path failed stops using it (its peers then run into the short
        time limit once and stop as well)."""
        err = None
        t_ipc = t_pg = float("inf")

        def note(msg):
            nonlocal err
            if err is None:
                err = msg

        def ipc_call(t):
            if err is None:
                try:
                    self.native.all_reduce_(t)
                except Exception as e:
                    note(repr(e))
            return t

        self.native.set_timeout(check_timeout_s)
        gen = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        for n in sizes:
            x = torch.randn(n, generator=gen).to(self.device)
            want = x.clone()
            self._group_all_reduce(want)
            for _ in range(3):  # both mailbox sets, and the same set twice
                got = ipc_call(x.clone())
            torch.cuda.synchronize(self.device)
            if self.native.timed_out():
                note("a peer did not arrive within the time limit")
            scale = float(want.abs().max().clamp_min(1e-6))
            if not bool(((got - want).abs().max() / scale) < 1e-5):
                note(f"mismatch against the process group at n={n}")
        x = torch.zeros(sizes[-1], device=self.device)
        for which in ("ipc", "pg"):
            fn = (lambda: ipc_call(x)) if which == "ipc" else (lambda: self._group_all_reduce(x))
            fn()
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize(self.device)
            dt = (time.perf_counter() - t0) / iters * 1e6
            if which == "ipc":
                t_ipc = dt
            else:
                t_pg = dt
        if self.native.timed_out():
            note("a peer did not arrive within the time limit")
        err = self._agree(err)
        if err is None:
            self.native.set_timeout(self.timeout_s)
        # rank 0's clock decides, so every rank takes the same path
        times = [(t_ipc, t_pg)] if self.rank == 0 else [None]
        src = tdist.get_global_rank(self.group, 0) if self.group is not None else 0
        tdist.broadcast_object_list(times, src=src, group=self.group)
        return err, times[0][0], times[0][1]


def _decide(group, device):
    mode = os.environ.get("PDT_SYNCBN_COMM", "auto").lower()
    info = {"mode": mode, "path": "group", "why": None, "ipc_us": None, "group_us": None}
    if mode in ("torch", "rccl", "group", "nccl") or device.type != "cuda":
        info["why"] = "selected" if device.type == "cuda" else "cpu tensors"
        return None, info
    if mode == "auto" and tdist.get_backend(group) != "nccl":
        # GPU tensors under a gloo group (the one-GPU rehearsal of W > 1).  Such a job stages every collective
        # through the host itself (dist.staged_through_host; profiles/r3_w2_hang.md has the stall that c10d's own
        # gloo staging ran into and what it depends on) and the host-staged path is the one to rehearse there.  The
        # IPC kernels are for RCCL jobs, where every dependency stays on the device; PDT_SYNCBN_COMM=ipc still
        # forces them.
        info["why"] = "the group is not an RCCL group"
        return None, info
    ipc = IpcSmallAllReduce(group, device)
    if not ipc.ok:
        info["why"] = ipc.error
        return None, info
    err, t_ipc, t_pg = ipc.validate()
    info["ipc_us"], info["group_us"] = t_ipc, t_pg
    if err is not None:
        info["why"] = err
        return None, info
    if mode == "ipc" or t_ipc <= t_pg:
        info["path"] = "ipc"
        return ipc, info
    info["why"] = "the process group was faster"
    return None, info


def small_all_reduce_path(group, device):
    """The (cached) decision for ``group``: ``(IpcSmallAllReduce or None, info dict)``.  Collective on first use."""
    key = (id(group) if group is not None else None, str(device))
    if key not in _paths:
        _paths[key] = _decide(group, device)
        if tdist.get_rank(group) == 0 and os.environ.get("PDT_QUIET", "0") == "0":
            _, info = _paths[key]
            print(f"[syncbn-comm] {info}", file=sys.stderr, flush=True)
    return _paths[key]


_TRACE = os.environ.get("PDT_TRACE_COLL", "0") == "1"  # one stderr line per small all-reduce (issue / done)
_trace_n = 0


def _all_reduce_group(t: torch.Tensor, group) -> None:
    from ..dist import all_reduce_  # host-staged for GPU tensors on a gloo group (dist.staged_through_host)

    all_reduce_(t, group=group)


def small_all_reduce(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM of a small fp32 vector over ``group`` on the current stream."""
    global _trace_n
    ipc, _ = small_all_reduce_path(group, t.device)
    if ipc is not None and t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= ipc.capacity:
        ipc.all_reduce_(t)
    elif _TRACE:
        _trace_n += 1
        r = tdist.get_rank()
        print(f"[coll r{r} {time.monotonic():.4f}] small#{_trace_n} n={t.numel()} issue", file=sys.stderr, flush=True)
        _all_reduce_group(t, group)
        print(f"[coll r{r} {time.monotonic():.4f}] small#{_trace_n} done", file=sys.stderr, flush=True)
    else:
        _all_reduce_group(t, group)
    return t


# ------------------------------------------------------------------------------------------- gradient buckets
class IpcBucketAllReduce(IpcSmallAllReduce):
    """Windows for the two-shot direct all-reduce of gradient buckets (``csrc/comm/ipc_allreduce.hip``, SURVEY 5.8
    layer 4): every rank stores shard j of its bucket straight into peer j's window (W-1 xGMI links at once), adds
    the W versions of its own shard in rank order and stores the result into every peer's window.  ``bucket_bytes``
    is the largest bucket it has to take; the windows cost 4x that per rank."""

    def __init__(self, group, device, bucket_bytes: int, timeout_s: float | None = None):
        world = tdist.get_world_size(group)
        words = (int(bucket_bytes) + 3) // 4
        cap = (words + world - 1) // world
        cap += cap & 1
        super().__init__(group, device, capacity=max(cap + 2, 1024), timeout_s=timeout_s)

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        self.native.all_reduce_large_(t)
        return t

    def validate(self, numel: int, iters: int = 5, check_timeout_s: float = 10.0):
        """-> (error or None, ipc_ms, group_ms) on fp32 and bf16 buffers of ``numel`` elements (random data) against the
        process group's all-reduce; collective, same discipline as :meth:`IpcSmallAllReduce.validate`."""
        err = None

        def note(msg):
            nonlocal err
            err = err or msg

        self.native.set_timeout(check_timeout_s)
        gen = torch.Generator().manual_seed(4321 + self.rank)
        t_ipc = t_pg = float("inf")
        numel -= numel % 2
        for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
            for n in (2, 4098, numel):
                x = torch.randn(n, generator=gen).to(self.device).to(dtype)
                want = x.float().clone()
                self._group_all_reduce(want)
                got = None
                for _ in range(2):
                    got = x.clone()
                    try:
                        self.native.all_reduce_large_(got)
                    except Exception as e:  # noqa: BLE001
                        note(repr(e))
                        break
                torch.cuda.synchronize(self.device)
                if self.native.timed_out():
                    note("a peer did not arrive within the time limit")
                if got is not None and err is None:
                    scale = float(want.abs().max().clamp_min(1e-6))
                    if not bool(((got.float() - want).abs().max() / scale) < tol):
                        note(f"mismatch against the process group at n={n}, {dtype}")
        if err is None:
            x = torch.zeros(numel, device=self.device)
            for which in ("ipc", "pg"):
                fn = (lambda: self.native.all_reduce_large_(x)) if which == "ipc" else (lambda: self._group_all_reduce(x))
                fn()
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize(self.device)
                dt = (time.perf_counter() - t0) / iters * 1e3
                if which == "ipc":
                    t_ipc = dt
                else:
                    t_pg = dt
            if self.native.timed_out():
                note("a peer did not arrive within the time limit")
        err = self._agree(err)
        if err is None:
            self.native.set_timeout(self.timeout_s)
        times = [(t_ipc, t_pg)] if self.rank == 0 else [None]
        src = tdist.get_global_rank(self.group, 0) if self.group is not None else 0
        tdist.broadcast_object_list(times, src=src, group=self.group)
        return err, times[0][0], times[0][1]


def bucket_all_reduce_path(group, device, bucket_bytes: int, mode: str | None = None):
    """Which path the gradient buckets take (``PDT_GRAD_COMM``): ``rccl`` (default) = the communicator's
    ``ncclAllReduce``; ``ipc`` = the two-shot direct kernel whenever its self-check against the process group
    passes; ``auto`` = the same check, then whichever was faster on a buffer of one bucket (rank 0's clock decides, so
    the job measures its own links at start-up).  -> ``(IpcBucketAllReduce or None, info dict)``; collective."""
    mode = (mode or os.environ.get("PDT_GRAD_COMM", "rccl")).lower()
    info = {"mode": mode, "path": "group", "why": None, "ipc_ms": None, "group_ms": None, "bucket_bytes": int(bucket_bytes)}
    if mode not in ("ipc", "auto") or device.type != "cuda" or tdist.get_world_size(group) == 1:
        info["why"] = "selected" if mode not in ("ipc", "auto") else "one rank / cpu tensors"
        return None, info
    ipc = IpcBucketAllReduce(group, device, bucket_bytes)
    if not ipc.ok:
        info["why"] = ipc.error
        return None, info
    err, t_ipc, t_pg = ipc.validate(int(bucket_bytes) // 4)
    info["ipc_ms"], info["group_ms"] = t_ipc, t_pg
    if err is not None:
        info["why"] = err
        ipc.close()
        return None, info
    if mode == "ipc" or t_ipc <= t_pg:
        info["path"] = "ipc"
        _paths[("bucket", id(group) if group is not None else None, str(device), id(ipc))] = (ipc, info)  # (close_all)
        return ipc, info
    info["why"] = "the process group was faster"
    ipc.close()
    return None, info


def reset_paths() -> None:
    _paths.clear()


def close_all() -> int:
    """Ordered teardown of every mailbox set this process created: ``close()`` each (synchronise, unmap the peers,
    free), then forget the per-group decisions.  -> number of IPC paths closed.  Part of ``dist.shutdown``, between
    the native communicators and ``destroy_process_group`` (every rank is behind the shutdown barrier by then)."""
    n = 0
    for ipc, _ in list(_paths.values()):
        if ipc is not None:
            ipc.close()
            n += 1
    _paths.clear()
    return n
