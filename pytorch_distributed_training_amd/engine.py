"""Training engine shared by ``main.py`` and ``bench.py``.

Builds the replica exactly in the reference's order
(``/root/reference/main.py:79-84``): criterion, optimizer (on the CPU
parameters), ``net.cuda()``, SyncBN conversion, DDP wrap, ``net.train()``; and
runs one step in the reference's order (``main.py:98-105``): H2D, forward,
loss, ``zero_grad`` (after forward), backward, step, loss reduce.

MI355X specifics: bf16 compute with fp32 master weights in a flat arena, input
cast + NCHW->NHWC fused in one kernel, SyncBN on its own communicator.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass

import torch
import torch.nn as tnn

from . import dist as pdist
from . import nn as pnn
from .models import build_model
from .ops import backend_for
from .optim import build_optimizer
from .parallel import DistributedDataParallel
from .utils import profiling


@dataclass
class EngineConfig:
    model: str = "resnet50"
    num_classes: int = 1000
    dtype: str = "bf16"  # compute dtype on GPU; CPU always computes fp32
    optimizer: str = "adam"
    lr: float = 1e-3
    momentum: float = 0.9
    weight_decay: float = 0.0
    sync_bn: bool = True
    bucket_cap_mb: float | None = None
    grad_dtype: str = "fp32"  # dtype of the gradient buckets (what the all-reduce moves): fp32, or bf16 = half the bytes
    broadcast_buffers: bool | str = "auto"
    reducer: str | None = None
    seed: int | None = None
    image_size: int | None = None
    # capture forward + backward + optimizer of the steady-state step in one hipGraph (launch-bound shapes:
    # the CIFAR-shaped reference config issues ~550 kernels of a few microseconds each per step)
    graph: bool = False
    graph_warmup: int = 3
    # SURVEY 5.2 debug mode: reducer launch-order / event assertions + a replica checksum every N steps
    check_replicas: bool = False
    check_replicas_every: int = 1


def compute_dtype(cfg: EngineConfig, device: torch.device) -> torch.dtype:
    if device.type != "cuda":
        return torch.float32
    return {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[cfg.dtype]


class CapturedStep:
    """One training step captured in a hipGraph: static input buffers, the graph, its static output.

    ``ipc`` is the SyncBN mailbox set (``parallel.ipc.IpcSmallAllReduce``) whose exchanges are INSIDE the captured
    kernels, or None.  Eager launches check the mailboxes' status word on the host at every launch
    (``IpcAllReduce::mailboxes_for``); a replayed graph launches nothing through that code, so ``replay`` reads the
    word itself, before it enqueues (what the previous replay left; no device sync): a peer that was late inside a
    replay fed this rank's BN kernels stale statistics, and the job must stop instead of training on them.  The
    kernels queued behind the exchange that timed out see the device-side copy of the word and do not wait again
    (``ipc_dead`` in ``csrc/comm/ipc_device.h``), so the host gets here one time limit after the peer went missing,
    not 106 of them."""

    def __init__(self, fn, inputs, stream: torch.cuda.Stream, ipc=None):
        self.inputs = tuple(inputs)
        # one mailbox set or several (SyncBN statistics; the gradient buckets' windows under --grad_comm ipc)
        self.ipcs = [i for i in (ipc if isinstance(ipc, (list, tuple)) else [ipc]) if i is not None]
        self.ipc = self.ipcs[0] if self.ipcs else None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=stream):
            self.output = fn(*self.inputs)

    def check_peers(self, when: str = "the previous replayed step") -> None:
        for ipc in self.ipcs:
            if ipc.timed_out():
                what = "gradient bucket all-reduce" if hasattr(ipc.native, "large_capacity_words") and \
                    type(ipc).__name__ == "IpcBucketAllReduce" else "SyncBatchNorm exchange"
                raise RuntimeError(f"{what}: a peer did not arrive within {ipc.timeout_s:.0f} s in {when} (IPC mailbox "
                                   f"status word set); what it left behind is not to be trusted")

    def replay(self, *inputs):
        self.check_peers()
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.output


class Engine:
    def __init__(self, cfg: EngineConfig, device: torch.device | None = None):
        env = pdist.env()
        self.cfg = cfg
        self.device = device if device is not None else env.device
        self.rank, self.world_size = env.rank, env.world_size
        if cfg.seed is not None:
            torch.manual_seed(cfg.seed + self.rank)
        kw = {}
        if cfg.image_size is not None and cfg.model.startswith("vit"):
            kw["image_size"] = cfg.image_size
        net = build_model(cfg.model, cfg.num_classes, **kw)
        # reference order: criterion + optimizer BEFORE .cuda() / SyncBN conversion / DDP (main.py:79-83)
        self.criterion = pnn.CrossEntropyLoss()
        self.optimizer = build_optimizer(cfg.optimizer, net.parameters(), cfg.lr, cfg.momentum, cfg.weight_decay)
        net = net.to(self.device)
        self.sync_bn_comm = None
        self._ipc = None  # the SyncBN mailboxes (parallel/ipc.py) when the in-kernel exchange is in use
        if cfg.sync_bn:
            sync_group = pnn.make_sync_bn_group()
            net = pnn.convert_sync_batchnorm(net, process_group=sync_group)
            if sync_group is not None and self.device.type == "cuda":
                # decide the statistics all-reduce path now (collective), not inside the first forward
                from .parallel.ipc import small_all_reduce_path

                self._ipc, self.sync_bn_comm = small_all_reduce_path(sync_group, self.device)
        self.dtype = compute_dtype(cfg, self.device)
        dev_ids = [self.device.index] if self.device.type == "cuda" else None
        self.net = DistributedDataParallel(
            net, device_ids=dev_ids, output_device=dev_ids[0] if dev_ids else None,
            bucket_cap_mb=cfg.bucket_cap_mb, broadcast_buffers=cfg.broadcast_buffers,
            grad_dtype={"fp32": None, "bf16": torch.bfloat16}[cfg.grad_dtype],
            compute_dtype=self.dtype, reducer=cfg.reducer, check_replicas=cfg.check_replicas)
        self._check_every = max(1, int(os.environ.get("PDT_DEBUG_REPLICAS_EVERY", cfg.check_replicas_every)))
        self._steps = 0
        if profiling.roctx_enabled():
            self.net.reducer.set_roctx(True)
        self.net.train()
        self._graph = None          # CapturedStep
        self._graph_key = None
        self._eager_steps = 0
        self._graph_failed = None
        self._capture_stream = None
        if cfg.graph and self.device.type == "cuda" and hasattr(self.optimizer, "set_capturable"):
            self.optimizer.set_capturable(True)

    # ------------------------------------------------------------------ data
    def prepare_input(self, imgs: torch.Tensor) -> torch.Tensor:
        """fp32 NCHW (or uint8 NHWC) batch -> compute dtype, channels_last on GPU."""
        if imgs.device != self.device:
            imgs = imgs.to(self.device, non_blocking=True)
        if self.device.type == "cuda":
            if imgs.dtype == self.dtype and imgs.is_contiguous(memory_format=torch.channels_last):
                return imgs
            scale = 1.0 / 255.0 if imgs.dtype == torch.uint8 else 1.0
            return backend_for(imgs).nchw_to_nhwc_lowp(imgs, self.dtype, scale)
        if imgs.dtype == torch.uint8:  # NHWC block from the loader -> ToTensor's fp32 NCHW in [0, 1]
            return imgs.permute(0, 3, 1, 2).float() / 255.0
        return imgs

    # ------------------------------------------------------------------ step
    def train_step(self, imgs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """One iteration in the reference's order; returns the local loss tensor
        (fp32 scalar on the device, not synchronised with the host).  Under ``graph=True`` this is the replayed
        graph's static output buffer: read or clone it before the next step overwrites it."""
        if self.cfg.graph and self.device.type == "cuda" and self._graph_failed is None:
            loss = self._graph_step(imgs, labels)
        else:
            loss = self._eager_step(imgs, labels)
        self._steps += 1
        if self.net.check_replicas and self._steps % self._check_every == 0:
            self.net.assert_replicas_equal()  # raises on the first diverged step instead of training on garbage
        return loss

    def device_only_step(self) -> bool:
        """True when every cross-rank exchange of a training step runs inside this framework's own kernels (SyncBN
        statistics inside the BN kernels, gradient buckets through the two-shot IPC all-reduce, no per-forward buffer
        broadcast): the multi-rank configuration whose hipGraph capture is verified against eager."""
        if self.world_size == 1:
            return True
        bn_ok = (not self.cfg.sync_bn) or self._ipc is not None
        return bool(bn_ok and getattr(self.net.reducer, "bucket_ipc", None) is not None and not self.net.broadcast_buffers)

    def close(self) -> None:
        """Ordered teardown of the replica (call before ``dist.shutdown``)."""
        g, self._graph = self._graph, None
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if g is not None:
            g.check_peers("the last replayed step")
        self.net.close()

    def _graph_step(self, imgs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """Replay the captured step when the batch has the captured shape; run eagerly otherwise (warm-up
        steps: reducer rebuild, lazy optimizer state, kernel attributes; and the short last batch of an epoch)."""
        key = (tuple(imgs.shape), imgs.dtype, tuple(labels.shape), labels.dtype)
        if self._graph is not None and key == self._graph_key:
            return self._graph.replay(imgs, labels)
        # capture only a steady-state step: past the warm-up, gradients synchronised, and the reducer's one-off bucket
        # rebuild (start of iteration 2) behind us, so the captured launch sequence is the one every later step has
        if (self._graph is None and self._eager_steps >= self.cfg.graph_warmup and self.net._require_sync
                and getattr(self.net.reducer, "rebuilt", True)):
            try:
                self._capture(imgs, labels, key)
                return self._graph.output
            except Exception as e:  # leave the job running on the eager path and say why
                self._graph_failed = repr(e)
                self._graph = None
                torch.cuda.synchronize(self.device)
                if self.rank == 0:
                    print(f"[pdt] hipGraph capture failed, continuing eagerly: {self._graph_failed}", file=sys.stderr, flush=True)
        return self._eager_step(imgs, labels)

    def _capture(self, imgs, labels, key) -> None:
        s_imgs = imgs.to(self.device, copy=True)
        s_labels = labels.to(self.device, copy=True)
        torch.cuda.synchronize(self.device)
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._capture_stream):
            # autograd runs an AccumulateGrad node on the stream it was created on: the reducer keeps those
            # nodes alive, so re-create them on the capture stream (and free the ones the warm-up made)
            for p in self.net.parameters():
                p.grad = None
            if hasattr(self.net.reducer, "rehook"):
                self.net.reducer.rehook()
        self._graph = CapturedStep(self._eager_step, (s_imgs, s_labels), self._capture_stream,
                                   [self._ipc, getattr(self.net.reducer, "bucket_ipc", None)])
        self._graph_key = key
        # the capture pass enqueued nothing: run the step it recorded for this batch
        self._graph.graph.replay()

    def _eager_step(self, imgs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        self._eager_steps += 1
        with profiling.range_("h2d+cast"):
            x = self.prepare_input(imgs)
            labels = labels.to(self.device, non_blocking=True)
        with profiling.range_("forward"):
            output = self.net(x)
            loss = self.criterion(output, labels)
        self.optimizer.zero_grad()  # after forward, before backward (main.py:102)
        with profiling.range_("backward"):
            loss.backward()
        with profiling.range_("optimizer"):
            self.optimizer.step()
        return loss.detach()

    @torch.no_grad()
    def evaluate(self, loader) -> float:
        """Top-1 accuracy (the reference's commented-out loop, main.py:119-130),
        sharded by whatever sampler the loader uses and summed across ranks."""
        self.net.eval()
        cnt = torch.zeros(2, dtype=torch.float64, device=self.device)
        for imgs, labels in loader:
            out = self.net(self.prepare_input(imgs))
            labels = labels.to(self.device)
            cnt[0] += (out.argmax(1) == labels).sum()
            cnt[1] += labels.numel()
        if pdist.is_initialized() and self.world_size > 1:
            cnt = cnt.float() if self.device.type == "cuda" else cnt
            pdist.all_reduce_(cnt)
        self.net.train()
        return float(cnt[0].item() / max(cnt[1].item(), 1.0))
