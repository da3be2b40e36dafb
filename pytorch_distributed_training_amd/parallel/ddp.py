"""DistributedDataParallel: one process per GPU, bucketed overlapped all-reduce.

Reference call: ``DDP(net, device_ids=[local_rank], output_device=local_rank)``
(``/root/reference/main.py:83``).  What stock DDP does under that call, and
what this wrapper does instead:

====================  ======================================  =====================================
step                  stock DDP                               here
====================  ======================================  =====================================
verify replicas       all-gather of 161 shape records (S1)    one 8-byte hash all-gather
initial state sync    coalesced broadcasts <= 250 MiB (S2)    params/buffers flattened into arenas,
                                                              ONE broadcast per arena
per-forward buffers   broadcast of 159 buffers (S3)           ``broadcast_buffers="auto"``: skipped
                                                              when every BN is SyncBN (replicas
                                                              already compute identical stats)
gradient reduction    c10d Reducer, 25 MiB NCCL buckets (S5)  our reducer (C++ on GPU / Python spec
                                                              on CPU), xGMI-sized buckets, grads
                                                              alias the flat bucket
bucket rebuild        broadcast of index lists (S6)           deterministic + 8-byte hash check
====================  ======================================  =====================================
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as tdist
import torch.nn as tnn

from ..utils.env import native
from .arena import BufferArena, ParamArena
from .buckets import DEFAULT_BUCKET_CAP_MB, DEFAULT_FIRST_BUCKET_MB
from .comm import TorchComm
from .reducer_py import PyReducer


def _param_signature_hash(module: tnn.Module) -> int:
    h = 1469598103934665603
    for n, p in module.named_parameters():
        for tok in (n, str(tuple(p.shape)), str(p.dtype)):
            for ch in tok.encode():
                h = ((h ^ ch) * 1099511628211) & 0x7FFFFFFFFFFFFFFF
    return h


class DistributedDataParallel(tnn.Module):
    def __init__(self, module: tnn.Module, device_ids=None, output_device=None, process_group=None,
                 bucket_cap_mb: float | str | None = None, first_bucket_mb: float | None = None,
                 broadcast_buffers: bool | str = "auto", gradient_as_bucket_view: bool = True,
                 grad_dtype: torch.dtype | None = None, compute_dtype: torch.dtype | None = None,
                 reducer: str | None = None, find_unused_parameters: bool = False, check_replicas: bool = False):
        super().__init__()
        self.module = module
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("DistributedDataParallel needs a module with parameters")
        devs = {p.device for p in params}
        if len(devs) != 1:
            raise ValueError(f"module parameters span devices {devs}; one process drives exactly one device")
        self.device = params[0].device
        if device_ids is not None:
            if len(device_ids) != 1:
                raise ValueError("single-process multi-GPU is not supported: pass exactly one device id")
            want = torch.device("cuda", int(device_ids[0]) if not isinstance(device_ids[0], torch.device)
                                else device_ids[0].index)
            if self.device.type == "cuda" and self.device != want:
                raise ValueError(f"module is on {self.device} but device_ids={device_ids}")
        self.output_device = output_device
        self.process_group = process_group
        self.comm = TorchComm(process_group)
        self.world_size, self.rank = self.comm.world_size, self.comm.rank
        self.check_replicas = check_replicas or os.environ.get("PDT_DEBUG_REPLICAS", "0") == "1"
        self.find_unused_parameters = find_unused_parameters

        # S1: replicas must have the same parameter list
        hashes = self.comm.all_gather_i64(_param_signature_hash(module), self.device)
        if len(set(hashes)) != 1:
            raise RuntimeError(f"parameter names/shapes/dtypes differ across ranks (hashes {hashes})")

        # flatten + S2: one broadcast per arena
        self.param_arena = ParamArena(module)
        self.buffer_arena = BufferArena(module)
        for flat in self.param_arena.flat_tensors() + self.buffer_arena.flat_tensors():
            self.comm.broadcast(flat, 0)
        if compute_dtype is not None and compute_dtype != torch.float32:
            self.param_arena.attach_lowp(compute_dtype)

        from ..nn.modules import SyncBatchNorm

        bns = [m for m in module.modules() if isinstance(m, tnn.modules.batchnorm._BatchNorm)]
        # replicas compute identical running statistics only when EVERY BN is a SyncBN (from the module scan, not
        # from the argument: plain BN with broadcast_buffers=False legitimately keeps per-rank statistics)
        self._all_sync_bn = all(isinstance(m, (SyncBatchNorm, tnn.SyncBatchNorm)) for m in bns)
        if broadcast_buffers == "auto":
            broadcast_buffers = not self._all_sync_bn
        self.broadcast_buffers = bool(broadcast_buffers) and self.world_size > 1

        self.bucket_autotune = None
        if isinstance(bucket_cap_mb, str):
            if bucket_cap_mb.lower() != "auto":
                bucket_cap_mb = float(bucket_cap_mb)
            else:
                bucket_cap_mb = self._autotune_bucket_cap(grad_dtype or torch.float32)
        cap = int((bucket_cap_mb if bucket_cap_mb is not None else
                   float(os.environ.get("PDT_BUCKET_CAP_MB", DEFAULT_BUCKET_CAP_MB))) * (1 << 20))
        first = int((first_bucket_mb if first_bucket_mb is not None else
                     float(os.environ.get("PDT_FIRST_BUCKET_MB", DEFAULT_FIRST_BUCKET_MB))) * (1 << 20))
        names = [n for n, _ in module.named_parameters()]
        kind = (reducer or os.environ.get("PDT_REDUCER", "auto")).lower()
        ext = native()
        if kind == "auto":
            kind = "native" if (ext is not None and hasattr(ext, "Reducer")) else "py"
        if kind == "native":
            if ext is None or not hasattr(ext, "Reducer"):
                raise RuntimeError("native reducer requested but the _C extension is not built")
            from .reducer_native import NativeReducer
            self.reducer = NativeReducer(self.param_arena.params, self.comm, cap, first, grad_dtype,
                                         gradient_as_bucket_view, names, process_group)
        else:
            self.reducer = PyReducer(self.param_arena.params, self.comm, cap, first, grad_dtype,
                                     gradient_as_bucket_view, names)
        self.reducer_kind = kind
        self._require_sync = True
        if self.check_replicas or os.environ.get("PDT_DEBUG_REDUCER", "0") == "1":
            self.reducer.set_debug(True)

    def _autotune_bucket_cap(self, dtype) -> float:
        """``bucket_cap_mb="auto"``: time the all-reduce of a few bucket sizes on THIS job's links (xGMI ring
        steps are per-link bound, a PCIe / multi-node fabric behaves differently) and keep the smallest size
        within 10 % of the best bus bandwidth: smaller buckets start the overlap earlier in backward.  Rank 0's
        pick is broadcast so every rank plans identical buckets."""
        from .buckets import DEFAULT_BUCKET_CAP_MB, autotune_bucket_cap

        if self.world_size == 1:
            return float(DEFAULT_BUCKET_CAP_MB)

        def timed(t):
            self.comm.all_reduce(t)
            if t.is_cuda:
                torch.cuda.synchronize(t.device)

        _, res = autotune_bucket_cap(timed, self.device, dtype, candidates_mb=(4, 8, 16, 32, 64))
        best_bw = max(mb / s for mb, s in res.items())
        pick = min(mb for mb, s in res.items() if mb / s >= 0.9 * best_bw)
        t = torch.tensor([float(pick)], dtype=torch.float32, device=self.device)
        self.comm.broadcast(t, 0)
        self.bucket_autotune = {"picked_mb": float(t.item()), "seconds_per_allreduce": {str(k): v for k, v in res.items()}}
        return float(t.item())

    # ------------------------------------------------------------------ API
    def forward(self, *inputs, **kwargs):
        if self.broadcast_buffers and self.training:
            for flat in self.buffer_arena.flat_tensors():  # S3 (one collective per dtype arena)
                self.comm.broadcast(flat, 0)
        out = self.module(*inputs, **kwargs)
        if torch.is_grad_enabled() and self._require_sync:
            self.reducer.prepare_for_backward()
        return out

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: skip the all-reduce inside the context."""
        old = self._require_sync
        self._require_sync = False
        self.reducer.require_sync = False
        try:
            yield
        finally:
            self._require_sync = old
            self.reducer.require_sync = old

    def state_dict(self, *args, **kwargs):
        """``module.``-prefixed keys like stock DDP; checkpoints use ``self.module.state_dict()``."""
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        if assign:
            raise ValueError("assign=True would detach parameters from the flat arena")
        r = super().load_state_dict(state_dict, strict=strict)
        self.param_arena.refresh_lowp()
        return r

    @torch.no_grad()
    def assert_replicas_equal(self) -> None:
        """Debug aid: every rank must hold bit-identical parameters (and buffers when they are kept
        in sync).  One 8-byte all-gather per arena; ``check_replicas=True`` / ``PDT_DEBUG_REPLICAS=1`` makes the
        engine call it every ``PDT_DEBUG_REPLICAS_EVERY`` (default 1) steps."""
        arenas = [("parameters", f) for f in self.param_arena.flat_tensors()]
        if self.broadcast_buffers or self._all_sync_bn:
            arenas += [("buffers", f) for f in self.buffer_arena.flat_tensors()]
        for what, flat in arenas:
            s = flat.double().sum().item()
            b = flat.contiguous().view(torch.uint8).to(torch.int64)
            # position-weighted byte sum: a swap of two values does not cancel
            w = (torch.arange(b.numel(), device=b.device, dtype=torch.int64) % 8191) + 1
            h = int((b * w).sum().item()) & 0x7FFFFFFFFFFFFFFF
            sums = self.comm.all_gather_i64(h, self.device)
            if len(set(sums)) != 1:
                raise AssertionError(f"replica divergence in the {what} arena ({flat.dtype}): checksums {sums} "
                                     f"(rank {self.rank}, sum {s})")

    def close(self) -> None:
        """Ordered teardown: hooks off, the reducer's communicator drained and destroyed."""
        self.reducer.close()

    def __del__(self):
        try:
            self.reducer.remove_hooks()
        except Exception:
            pass
