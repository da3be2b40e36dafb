"""Modules: parameter containers with torch-compatible ``state_dict`` keys whose
``forward`` runs the fused HIP nodes of :mod:`.functional`.

* ``Conv2d`` / ``Linear``: subclasses of the torch modules (same init, same keys).
* ``BatchNorm2d``: ``forward(x, residual=None, relu=False, stats=None)``.
* ``SyncBatchNorm`` + ``convert_sync_batchnorm``: the reference converts every
  BN after building the optimizer (``/root/reference/main.py:80,82``), so the
  conversion re-binds the *same* ``weight``/``bias`` Parameters and buffers
  (SURVEY §4.2); unlike the stock module it also runs on CPU tensors over gloo.
* ``CrossEntropyLoss``: fused softmax-CE (``main.py:79,101``).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as tdist
import torch.nn as tnn

from . import functional as F
from ..ops import backend_for


class Conv2d(tnn.Conv2d):
    def forward(self, x):
        if (self.bias is None and self.groups == 1 and self.dilation == (1, 1) and isinstance(self.padding, tuple)
                and self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1]
                and self.kernel_size[0] == self.kernel_size[1]):
            return F.conv2d(x, self.weight, self.stride[0], self.padding[0])[0]
        w = self.weight
        lp = getattr(w, "_lp", None)
        if lp is not None and lp.dtype == x.dtype:
            # differentiable through the master weight: d(lp)/d(w) = identity
            w = w + (lp - w).detach() if w.dtype == x.dtype else w.to(x.dtype)
        elif w.dtype != x.dtype:
            w = w.to(x.dtype)
        b = None if self.bias is None else self.bias.to(x.dtype)
        return self._conv_forward(x, w, b)


class Linear(tnn.Linear):
    """``forward(x, residual=None)``: ``x @ W^T + b (+ residual)`` on the own MFMA GEMM (csrc/kernels/gemm.hip) with
    bias and residual in its epilogue for bf16 GPU tensors (fp32 master weight, bf16 working copy ``weight._lp``);
    the PyTorch oracle on CPU tensors."""

    def forward(self, x, residual=None):
        return F.linear(x, self.weight, self.bias, residual)


class BatchNorm2d(tnn.BatchNorm2d):
    _sync = False
    process_group = None

    def _cfg(self) -> F.BNConfig:
        tracked = self.running_mean is not None
        use_batch = self.training or not tracked  # batch statistics vs running statistics
        return F.BNConfig(self.running_mean if tracked else None, self.running_var if tracked else None,
                          self.num_batches_tracked if (self.training and tracked) else None,
                          self.momentum, self.eps, use_batch, sync=self._sync and self.training,
                          group=self.process_group)

    def forward(self, x, residual=None, relu: bool = False, stats=None, lazy_residual_grad: bool = False,
                defer_apply: bool = False):
        if x.dim() != 4:
            raise ValueError(f"expected 4D input (got {x.dim()}D input)")
        cfg = self._cfg()
        if not cfg.training:
            stats = None
        return F.bn_act(x, self.weight, self.bias, cfg, residual=residual, relu=relu, stats=stats,
                        lazy_residual_grad=lazy_residual_grad, defer_apply=defer_apply)


class SyncBatchNorm(BatchNorm2d):
    """Cross-replica BN (``main.py:82``).  Synchronises only when training and the
    group has more than one rank; otherwise it is exactly ``BatchNorm2d``."""

    _sync = True

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None, device=None, dtype=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats, device=device, dtype=dtype)
        self.process_group = process_group

    @classmethod
    def convert_sync_batchnorm(cls, module: tnn.Module, process_group=None) -> tnn.Module:
        return convert_sync_batchnorm(module, process_group)


def convert_sync_batchnorm(module: tnn.Module, process_group=None) -> tnn.Module:
    """Recursively replace every ``BatchNorm*`` by ``SyncBatchNorm`` re-using the
    same Parameter / buffer objects (an optimizer built earlier stays valid)."""
    out = module
    if isinstance(module, tnn.modules.batchnorm._BatchNorm) and not isinstance(module, SyncBatchNorm):
        out = SyncBatchNorm(module.num_features, module.eps, module.momentum, module.affine,
                            module.track_running_stats, process_group)
        if module.affine:
            with torch.no_grad():
                out.weight = module.weight
                out.bias = module.bias
        out.running_mean = module.running_mean
        out.running_var = module.running_var
        out.num_batches_tracked = module.num_batches_tracked
        out.training = module.training
        if hasattr(module, "qconfig"):
            out.qconfig = module.qconfig
    for name, child in module.named_children():
        out.add_module(name, convert_sync_batchnorm(child, process_group))
    del module
    return out


def make_sync_bn_group():
    """A dedicated communicator for SyncBN so its tiny latency-bound all-reduces
    never queue behind a gradient bucket on the same RCCL stream."""
    if not (tdist.is_available() and tdist.is_initialized()) or tdist.get_world_size() == 1:
        return None
    if os.environ.get("PDT_SYNCBN_GROUP", "own").lower() == "default":  # diagnostic: share the WORLD group
        return None
    return tdist.new_group(ranks=list(range(tdist.get_world_size())))


class ReLU(tnn.ReLU):
    pass


class MaxPool3x3s2(tnn.Module):
    """The ResNet stem pool: 3x3 / stride 2 / pad 1."""

    def forward(self, x):
        return F.maxpool3x3s2(x)


class GlobalAvgPool(tnn.Module):
    """``AdaptiveAvgPool2d(1)`` + ``flatten(1)``."""

    def forward(self, x):
        return F.global_avgpool(x)


class CrossEntropyLoss(tnn.Module):
    """Mean-reduced cross entropy over class logits (``main.py:79``)."""

    def forward(self, logits, labels):
        return F.softmax_cross_entropy(logits, labels)
