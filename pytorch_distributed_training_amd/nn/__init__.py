"""Modules and fused autograd nodes (reference ``main.py:79-82``)."""
from . import functional
from .modules import (BatchNorm2d, Conv2d, CrossEntropyLoss, GlobalAvgPool, Linear, MaxPool3x3s2, ReLU,
                      SyncBatchNorm, convert_sync_batchnorm, make_sync_bn_group)

__all__ = [
    "functional", "Conv2d", "Linear", "BatchNorm2d", "SyncBatchNorm", "convert_sync_batchnorm",
    "make_sync_bn_group", "ReLU", "MaxPool3x3s2", "GlobalAvgPool", "CrossEntropyLoss",
]
