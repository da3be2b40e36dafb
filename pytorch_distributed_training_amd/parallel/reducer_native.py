"""Python face of the C++ reducer (``csrc/reducer``) and the native RCCL communicator.

The hot path (hooks, bucket flatten launch, stream fencing, ``ncclAllReduce``,
finalize) is C++; this wrapper only does the one-time jobs that need Python
objects: exchanging the RCCL unique id through the c10d store, and the
hash-check + rebuild between iteration 1 and 2.
"""
from __future__ import annotations

import itertools
import os
import sys

import torch
import torch.distributed as tdist

from ..utils.env import require_native
from .buckets import order_hash

_comm_counter = itertools.count()
_shared_comm = {}


def make_rccl_comm(group=None, high_priority: bool = True):
    """Create (or reuse) the native communicator for ``group`` (default: WORLD).

    Rank 0 calls ``ncclGetUniqueId`` and publishes it in the process-group store;
    every rank then joins with ``ncclCommInitRank``.
    """
    C = require_native()
    key = (id(group), high_priority)
    if key in _shared_comm:
        return _shared_comm[key]
    if not (tdist.is_available() and tdist.is_initialized()) or tdist.get_world_size(group) == 1:
        if os.environ.get("PDT_FORCE_COMM", "0") != "1":
            return None
        # size-1 communicator: lets one GPU run the real flatten -> fence -> ncclAllReduce -> event path
        comm = C.RcclComm(0, 1, C.RcclComm.make_unique_id(), torch.cuda.current_device(), high_priority)
        _shared_comm[key] = comm
        return comm
    rank, world = tdist.get_rank(group), tdist.get_world_size(group)
    store = tdist.distributed_c10d._get_default_store()
    name = f"pdt/rccl_uid/{next(_comm_counter)}"
    if rank == 0:
        store.set(name, C.RcclComm.make_unique_id())
    uid = bytes(store.get(name))
    comm = C.RcclComm(rank, world, uid, torch.cuda.current_device(), high_priority)
    _shared_comm[key] = comm
    return comm


def _rccl_or_none(process_group, world: int, device):
    """The job's own communicator, or None (-> the reducer all-reduces through the c10d group instead) when
    ``PDT_COMM=torch`` asks for that or when ANY rank failed to create it: the ranks agree on the outcome
    through the c10d group, so a node whose RCCL bootstrap fails degrades the whole job, not one rank."""
    if os.environ.get("PDT_COMM", "native").lower() == "torch" and world > 1:
        return None
    if world == 1:
        return make_rccl_comm(process_group, high_priority=True)  # None unless PDT_FORCE_COMM=1
    if tdist.get_backend(process_group) != "nccl":
        return None  # GPU tensors under a gloo group (one-GPU rehearsal of W > 1): ranks may share a device
    comm, err = None, None
    try:
        comm = make_rccl_comm(process_group, high_priority=True)
    except Exception as e:  # noqa: BLE001 - reported below, then every rank takes the c10d path
        err = e
    ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=device)
    tdist.all_reduce(ok, op=tdist.ReduceOp.MIN, group=process_group)
    if int(ok.item()) == 1:
        return comm
    if tdist.get_rank(process_group) == 0 or err is not None:
        print(f"[pdt] native RCCL communicator unavailable ({err!r}); all-reducing through torch.distributed",
              file=sys.stderr, flush=True)
    _shared_comm.pop((id(process_group), True), None)
    return None


def close_all_comms() -> None:
    """``RcclComm.close()`` on every communicator this process created (stop watchdog -> drain stream ->
    ncclCommDestroy); call after the last collective and before ``destroy_process_group``."""
    for comm in list(_shared_comm.values()):
        if comm is not None:
            comm.close()
    _shared_comm.clear()


class NativeReducer:
    def __init__(self, params, comm, bucket_cap_bytes: int, first_bucket_bytes: int, grad_dtype=None,
                 gradient_as_bucket_view: bool = True, names=None, process_group=None):
        C = require_native()
        self.params = list(params)
        self.comm = comm  # TorchComm: used for the one-off hash check only
        self.world = comm.world_size
        self.device = self.params[0].device
        names = list(names) if names is not None else [f"param{i}" for i in range(len(self.params))]
        self.rccl = None
        pg = None
        if self.device.type == "cuda":
            self.rccl = _rccl_or_none(process_group, self.world, self.device)
        if self.rccl is None and self.world > 1:  # gloo, or the c10d RCCL group when the own communicator is off
            pg = process_group if process_group is not None else tdist.distributed_c10d._get_default_group()
        self._r = C.Reducer(self.params, names, self.rccl, pg, self.world, int(bucket_cap_bytes),
                            int(first_bucket_bytes), grad_dtype, bool(gradient_as_bucket_view))
        self._require_sync = True
        # gradient buckets through the two-shot direct all-reduce over IPC windows instead of ncclAllReduce
        # (PDT_GRAD_COMM=ipc|auto; default rccl: the kernel has run between two processes on one GPU, never across
        # xGMI, and "auto" is the switch that lets a job measure both on its own links)
        self.bucket_comm = {"mode": os.environ.get("PDT_GRAD_COMM", "rccl").lower(), "path": "group"}
        self.bucket_ipc = None
        if self.device.type == "cuda" and self.world > 1 and self.bucket_comm["mode"] in ("ipc", "auto"):
            from .ipc import bucket_all_reduce_path

            self.bucket_ipc, self.bucket_comm = bucket_all_reduce_path(process_group, self.device, int(bucket_cap_bytes))
            if self.bucket_ipc is not None:
                self._r.set_bucket_ipc(self.bucket_ipc.native)
            if tdist.get_rank(process_group) == 0 and os.environ.get("PDT_QUIET", "0") == "0":
                print(f"[grad-comm] {self.bucket_comm}", file=sys.stderr, flush=True)

    def ipc_bucket_launches(self) -> int:
        return int(self._r.ipc_bucket_launches())

    # -- same surface as PyReducer ------------------------------------------
    @property
    def require_sync(self) -> bool:
        return self._require_sync

    @require_sync.setter
    def require_sync(self, v: bool) -> None:
        self._require_sync = bool(v)
        self._r.set_require_sync(bool(v))

    @property
    def rebuilt(self) -> bool:
        return self._r.rebuilt

    @property
    def iteration(self) -> int:
        return self._r.iteration

    def prepare_for_backward(self) -> None:
        if not self._r.rebuilt and self._r.iteration >= 1:
            self._rebuild()
        self._r.prepare_for_backward()

    def _rebuild(self) -> None:
        ready = list(self._r.ready_order())
        seen = set(ready)
        order = ready + [i for i in self._r.current_order() if i not in seen]
        hashes = self.comm.all_gather_i64(order_hash(order), self.device)
        if len(set(hashes)) != 1:  # ranks disagree (non-deterministic graph): adopt rank 0's order
            t = torch.tensor(order, dtype=torch.int64, device=self.device)
            self.comm.broadcast(t, 0)
            order = [int(v) for v in t.tolist()]
        self._r.rebuild(order)

    def bucket_specs(self):
        from .buckets import BucketSpec

        return [BucketSpec(list(idx), [], n) for idx, n in zip(self._r.bucket_indices(), self._r.bucket_numels())]

    def bucket_tensors(self):
        return self._r.bucket_tensors()

    def remove_hooks(self) -> None:
        self._r.remove_hooks()

    def rehook(self) -> None:
        """Recreate the AccumulateGrad nodes under the CURRENT stream (they remember the stream they were
        created on; a hipGraph capture must not pull the default stream in through them)."""
        self._r.install_hooks()

    # -- debug mode / markers -----------------------------------------------------------
    def set_debug(self, on: bool) -> None:
        """Bucket launch-order / readiness / completion-event assertions in the C++ reducer (also PDT_DEBUG_REDUCER=1)"""
        self._r.set_debug(bool(on))

    def set_roctx(self, on: bool) -> None:
        """one roctx range per bucket all-reduce (and around finalize), emitted from C++"""
        self._r.set_roctx(bool(on))

    def launch_log(self) -> list[int]:
        return list(self._r.launch_log())

    def debug_checks(self) -> int:
        return int(self._r.debug_checks())

    def close(self) -> None:
        """Ordered teardown of this reducer's side of the job: hooks off, communicator drained and destroyed."""
        self._r.remove_hooks()
        self._r.set_bucket_ipc(None)  # (the windows themselves are closed collectively: ipc.close_all in dist.shutdown)
        if self.rccl is not None:
            self.rccl.close()

    # -- per-bucket all-reduce timing ------------------------------------------------------
    def set_timing(self, on: bool) -> bool:
        """Timed events around every bucket's all-reduce; True when the timing is available (native RCCL
        communicator on a GPU).  Not to be left on while a hipGraph is captured."""
        self._r.set_timing(bool(on))
        return bool(on) and self.rccl is not None

    def comm_stats(self) -> dict:
        """Of the most recent iteration: per-bucket all-reduce time, how long the compute stream had to wait
        for the buckets after backward's own work ended, and the overlap fraction = 1 - exposed / total."""
        per, exposed, span = self._r.comm_stats()
        total = float(sum(per))
        sizes = [int(t.numel() * t.element_size()) for t in self._r.bucket_tensors()]
        return {"bucket_bytes": sizes, "bucket_allreduce_ms": [float(v) for v in per], "allreduce_ms_total": total,
                "allreduce_span_ms": float(span), "exposed_ms": float(exposed),
                "overlap_fraction": (1.0 - min(exposed, total) / total) if total > 0 else 1.0}
