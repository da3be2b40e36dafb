"""Collective interface used by the reducers.

Two implementations of one small surface, so the CPU/gloo tests exercise the
same reducer logic as the GPU path:

* :class:`TorchComm`  -- ``torch.distributed`` (gloo on CPU; RCCL through c10d on GPU).
* ``_C.RcclComm``     -- the native communicator (``ncclCommInitRank`` on a unique
  id exchanged through the c10d store, collectives enqueued on our own
  non-blocking HIP streams); used by the C++ reducer, see ``csrc/comm``.
"""
from __future__ import annotations

import torch
import torch.distributed as tdist


def _dist():
    from .. import dist

    return dist


class TorchComm:
    def __init__(self, group=None):
        self.group = group
        self.initialized = tdist.is_available() and tdist.is_initialized()
        self.world_size = tdist.get_world_size(group) if self.initialized else 1
        self.rank = tdist.get_rank(group) if self.initialized else 0

    def all_reduce_async(self, flat: torch.Tensor):
        if self.world_size == 1:
            return None
        if _dist().staged_through_host(flat, self.group):
            # GPU tensor on a gloo group: staged through the host by us (dist.staged_through_host says why); the
            # collective itself still runs asynchronously on the CPU copy
            host = flat.detach().cpu()
            return (tdist.all_reduce(host, group=self.group, async_op=True), host, flat)
        return tdist.all_reduce(flat, group=self.group, async_op=True)

    @staticmethod
    def wait(handle) -> None:
        if isinstance(handle, tuple):
            work, host, flat = handle
            work.wait()
            flat.copy_(host)
        elif handle is not None:
            handle.wait()

    def all_reduce(self, flat: torch.Tensor) -> None:
        if self.world_size > 1:
            _dist().all_reduce_(flat, group=self.group)

    def broadcast(self, flat: torch.Tensor, src: int = 0) -> None:
        if self.world_size > 1:
            _dist().broadcast_(flat, src=src, group=self.group)

    def all_gather_i64(self, value: int, device) -> list[int]:
        if self.world_size == 1:
            return [int(value)]
        if torch.device(device).type == "cuda" and tdist.get_backend(self.group) != "nccl":
            device = "cpu"  # (8 bytes: no reason to involve the GPU under gloo)
        t = torch.tensor([value], dtype=torch.int64, device=device)
        out = [torch.empty_like(t) for _ in range(self.world_size)]
        tdist.all_gather(out, t, group=self.group)
        return [int(o.item()) for o in out]
