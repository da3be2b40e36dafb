"""Python gradient reducer: the executable specification of the C++ reducer.

Same architecture as ``csrc/reducer`` (bucket plan, per-parameter
post-accumulate hooks, in-order bucket launch, rebuild from the observed
ready order after iteration 1, finalize callback) but with the collectives
going through :class:`..parallel.comm.TorchComm`.  It is the path the CPU/gloo
plumbing config runs (BASELINE.json config 1) and the oracle the C++ reducer
is tested against.  Behaviour being reproduced: SURVEY §2.2 C8 / §3.4
(``/root/reference/main.py:83,103``).

Invariants:
* a bucket is reduced only after every gradient in it has been written;
* buckets launch in plan order on every rank (bucket k waits for k-1), so the
  collective sequence is identical across ranks;
* the optimizer sees ``.grad`` only after every bucket's all-reduce completed;
* with ``gradient_as_bucket_view`` ``.grad`` aliases the flat bucket, and the
  reference's ``zero_grad()`` between forward and backward (``main.py:102``;
  ``set_to_none=True``) just drops the alias: the next backward re-creates it.
"""
from __future__ import annotations

import time

import torch

from ..ops import backend_for
from .buckets import BucketSpec, initial_order, order_hash, plan_buckets


class _Bucket:
    def __init__(self, spec: BucketSpec, params, dtype, device):
        self.spec = spec
        self.flat = torch.zeros(spec.numel, dtype=dtype, device=device)
        self.views = [self.flat[o: o + params[i].numel()] for i, o in zip(spec.param_indices, spec.offsets)]
        self.pending = len(spec.param_indices)
        self.ready = [False] * len(spec.param_indices)
        self.launched = False
        self.handle = None
        self.t_launch = 0.0


class PyReducer:
    def __init__(self, params, comm, bucket_cap_bytes: int, first_bucket_bytes: int,
                 grad_dtype: torch.dtype | None = None, gradient_as_bucket_view: bool = True, names=None):
        self.params = list(params)
        self.names = list(names) if names is not None else [f"param{i}" for i in range(len(self.params))]
        self.comm = comm
        self.world = comm.world_size
        self.cap, self.first_cap = int(bucket_cap_bytes), int(first_bucket_bytes)
        self.device = self.params[0].device
        self.grad_dtype = grad_dtype or self.params[0].dtype
        self.as_view = gradient_as_bucket_view
        self.numels = [p.numel() for p in self.params]
        self.require_sync = True
        self.iteration = 0
        self.rebuilt = False
        self._ready_order: list[int] = []
        self._callback_queued = False
        self._next_bucket = 0
        self._build(initial_order(len(self.params)))
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                       for i, p in enumerate(self.params) if p.requires_grad]
        self.last_stats: dict = {}
        self.debug = False
        self._launch_log: list[int] = []
        self._debug_checks = 0

    # same debug surface as the C++ reducer
    def set_debug(self, on: bool) -> None:
        self.debug = bool(on)

    def set_roctx(self, on: bool) -> None:  # markers come from the C++ reducer only
        pass

    def launch_log(self) -> list[int]:
        return list(self._launch_log)

    def debug_checks(self) -> int:
        return self._debug_checks

    def close(self) -> None:
        self.remove_hooks()

    # ------------------------------------------------------------- planning
    def _build(self, order: list[int]) -> None:
        order = [i for i in order if self.params[i].requires_grad]
        elem = torch.empty((), dtype=self.grad_dtype).element_size()
        specs = plan_buckets(self.numels, order, elem, self.cap, self.first_cap)
        self.buckets = [_Bucket(s, self.params, self.grad_dtype, self.device) for s in specs]
        self.where: dict[int, tuple[int, int]] = {}
        for b, bk in enumerate(self.buckets):
            for j, i in enumerate(bk.spec.param_indices):
                self.where[i] = (b, j)
        self.order = order

    def bucket_specs(self) -> list[BucketSpec]:
        return [b.spec for b in self.buckets]

    # ---------------------------------------------------------------- hooks
    def _make_hook(self, idx: int):
        def hook(param):
            self._on_grad_ready(idx)
        return hook

    def prepare_for_backward(self) -> None:
        for b in self.buckets:
            b.pending = len(b.spec.param_indices)
            b.ready = [False] * b.pending
            b.launched = False
            b.handle = None
        self._next_bucket = 0
        self._callback_queued = False
        self._launch_log = []
        if not self.rebuilt:
            self._ready_order = []

    def _on_grad_ready(self, idx: int) -> None:
        if not self.require_sync:  # inside no_sync(): gradients just accumulate locally
            return
        if not self._callback_queued:
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
            self._callback_queued = True
        if not self.rebuilt:
            self._ready_order.append(idx)
        b, j = self.where[idx]
        bk = self.buckets[b]
        if bk.ready[j]:
            raise RuntimeError(
                f"parameter {self.names[idx]} received a second gradient in one backward; call "
                f"prepare_for_backward() (DDP.forward) once per backward")
        bk.ready[j] = True
        bk.pending -= 1
        self._launch_ready()

    def _launch_ready(self) -> None:
        while self._next_bucket < len(self.buckets) and self.buckets[self._next_bucket].pending == 0:
            self._launch(self.buckets[self._next_bucket])
            self._next_bucket += 1

    def _launch(self, bk: _Bucket, from_finalize: bool = False) -> None:
        bidx = self.buckets.index(bk)
        if self.debug:
            assert bidx == len(self._launch_log), f"bucket {bidx} launched out of plan order"
            assert not bk.launched, f"bucket {bidx} launched twice in one iteration"
            assert from_finalize or (bk.pending == 0 and all(bk.ready)), f"bucket {bidx} launched with gradients missing"
            self._debug_checks += 1
        self._launch_log.append(bidx)
        grads, views = [], []
        for i, v in zip(bk.spec.param_indices, bk.views):
            g = self.params[i].grad
            if g is None:  # unused this iteration: contributes zeros
                v.zero_()
                continue
            grads.append(g)
            views.append(_shape_like(v, self.params[i]))  # slot keeps the parameter's physical layout
        scale = 1.0 / self.world
        be = backend_for(bk.flat)
        # one multi-tensor launch: copy + 1/W (+ cast) into the flat bucket (N2)
        be.flatten_scaled_views(grads, views, scale)
        bk.t_launch = time.perf_counter()
        bk.handle = self.comm.all_reduce_async(bk.flat)
        bk.launched = True

    # ------------------------------------------------------------- finalize
    def _finalize(self) -> None:
        self._callback_queued = False
        if not self.require_sync:
            return
        # parameters that never produced a gradient (unused in this graph) are zero-filled
        for bk in self.buckets[self._next_bucket:]:
            self._launch(bk, from_finalize=True)
        self._next_bucket = len(self.buckets)
        for bk in self.buckets:
            self.comm.wait(bk.handle)
            bk.handle = None
        if self.debug:
            assert all(bk.launched for bk in self.buckets) and len(self._launch_log) == len(self.buckets)
            self._debug_checks += 1
        for bk in self.buckets:
            for i, v in zip(bk.spec.param_indices, bk.views):
                p = self.params[i]
                if self.as_view and self.grad_dtype == p.dtype:
                    if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                        p.grad = _shape_like(v, p)
                else:
                    if p.grad is None:
                        p.grad = torch.empty_like(p)
                    p.grad.copy_(_shape_like(v, p))
        self.iteration += 1
        if not self.rebuilt:
            self._rebuild_from_ready_order()

    def _rebuild_from_ready_order(self) -> None:
        seen = set(self._ready_order)
        order = list(self._ready_order) + [i for i in self.order if i not in seen]
        # every rank must derive the same plan: compare a hash instead of broadcasting lists (S6)
        hashes = self.comm.all_gather_i64(order_hash(order), self.device)
        if len(set(hashes)) != 1:
            t = torch.tensor(order, dtype=torch.int64, device=self.device)
            self.comm.broadcast(t, 0)
            order = [int(v) for v in t.tolist()]
        old_grads = {i: self.params[i].grad for i in range(len(self.params))}
        self._build(order)
        self.rebuilt = True
        if self.as_view:
            # re-home the freshly reduced grads into the new buckets so .grad keeps aliasing a bucket
            for bk in self.buckets:
                for i, v in zip(bk.spec.param_indices, bk.views):
                    g = old_grads.get(i)
                    p = self.params[i]
                    if g is not None and self.grad_dtype == p.dtype:
                        nv = _shape_like(v, p)
                        nv.copy_(g)
                        p.grad = nv

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def _shape_like(flat_view: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """View a flat bucket slot with the parameter's logical shape AND physical layout."""
    if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
        co, ci, kh, kw = p.shape
        return flat_view.view(co, kh, kw, ci).permute(0, 3, 1, 2)
    return flat_view.view(p.shape)
