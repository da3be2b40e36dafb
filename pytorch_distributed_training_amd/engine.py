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
    broadcast_buffers: bool | str = "auto"
    reducer: str | None = None
    seed: int | None = None
    image_size: int | None = None


def compute_dtype(cfg: EngineConfig, device: torch.device) -> torch.dtype:
    if device.type != "cuda":
        return torch.float32
    return {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[cfg.dtype]


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
        if cfg.sync_bn:
            net = pnn.convert_sync_batchnorm(net, process_group=pnn.make_sync_bn_group())
        self.dtype = compute_dtype(cfg, self.device)
        dev_ids = [self.device.index] if self.device.type == "cuda" else None
        self.net = DistributedDataParallel(
            net, device_ids=dev_ids, output_device=dev_ids[0] if dev_ids else None,
            bucket_cap_mb=cfg.bucket_cap_mb, broadcast_buffers=cfg.broadcast_buffers,
            compute_dtype=self.dtype, reducer=cfg.reducer)
        self.net.train()

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
        return imgs.float() / 255.0 if imgs.dtype == torch.uint8 else imgs

    # ------------------------------------------------------------------ step
    def train_step(self, imgs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """One iteration in the reference's order; returns the local loss tensor
        (fp32 scalar on the device, not synchronised with the host)."""
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
            import torch.distributed as tdist

            cnt = cnt.float() if self.device.type == "cuda" else cnt
            tdist.all_reduce(cnt)
        self.net.train()
        return float(cnt[0].item() / max(cnt[1].item(), 1.0))
