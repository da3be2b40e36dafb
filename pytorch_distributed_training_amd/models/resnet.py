"""ResNet-18/34/50/101/152 (v1.5) with torchvision-compatible ``state_dict`` keys.

The reference builds ``torchvision.models.resnet50()`` (ImageNet ResNet-50
v1.5, 1000-way head, random init; ``/root/reference/main.py:40``).  torchvision
is not installed here and this is not a copy of it: the module tree is laid
out so that the parameter / buffer names (``conv1.weight``, ``bn1.*``,
``layer{1-4}.{i}.conv{1-3}.weight``, ``layer*.{i}.bn{1-3}.*``,
``layer*.0.downsample.{0,1}.*``, ``fc.{weight,bias}``), shapes, initialisers
(Kaiming-normal fan_out for convs, BN weight 1 / bias 0, default Linear init)
and the stride-on-3x3 placement match, so checkpoints interchange with
``strict=True``.  ResNet-50: 25 557 032 parameters in
161 tensors, 159 buffers.

Each ``conv -> BN -> (+identity) -> ReLU`` group runs as the two fused nodes of
:mod:`..nn.functional` (conv with BN-statistics epilogue; BN+add+ReLU).
"""
from __future__ import annotations

import os

import torch
import torch.nn as tnn

from .. import nn as pnn
from ..nn.functional import conv_bn_act, conv_bn_relu_maxpool

_LAZY_RES = os.environ.get("PDT_LAZY_RES_GRAD", "1") != "0"  # A/B knob for the gated residual gradient
_DEFER_DS = os.environ.get("PDT_DEFER_DS_BN", "1") != "0"    # A/B knob: downsample BN applied by its consumer
_COMPACT_DS = os.environ.get("PDT_COMPACT_DS_GRAD", "1") != "0"  # A/B knob: stride-2 downsample dgrad kept compact


def _conv3x3(cin, cout, stride=1):
    return pnn.Conv2d(cin, cout, 3, stride, 1, bias=False)


def _conv1x1(cin, cout, stride=1):
    return pnn.Conv2d(cin, cout, 1, stride, 0, bias=False)


class BasicBlock(tnn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = pnn.BatchNorm2d(planes)
        self.relu = pnn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = pnn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # the block input's second consumer is routed through conv1's node: that branch's gradient is
        # then added inside conv1's data-gradient epilogue instead of by a separate elementwise kernel
        out, identity = conv_bn_act(x, self.conv1, self.bn1, relu=True, return_skip=True)
        if self.downsample is not None:
            # the downsample BN's affine is applied by the block-output node while it reads the residual
            identity = conv_bn_act(identity, self.downsample[0], self.downsample[1], relu=False, defer_apply=_DEFER_DS,
                                   compact_skip_grad=_COMPACT_DS)
        # identity has exactly one consumer (this node): its gradient may come back ungated + mask
        return conv_bn_act(out, self.conv2, self.bn2, relu=True, residual=identity, lazy_residual_grad=_LAZY_RES)


class Bottleneck(tnn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv1x1(inplanes, planes)
        self.bn1 = pnn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride)  # v1.5: stride on the 3x3
        self.bn2 = pnn.BatchNorm2d(planes)
        self.conv3 = _conv1x1(planes, planes * self.expansion)
        self.bn3 = pnn.BatchNorm2d(planes * self.expansion)
        self.relu = pnn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # the block input's second consumer is routed through conv1's node: that branch's gradient is
        # then added inside conv1's data-gradient epilogue instead of by a separate elementwise kernel
        out, identity = conv_bn_act(x, self.conv1, self.bn1, relu=True, return_skip=True)
        if self.downsample is not None:
            # the downsample BN's affine is applied by the block-output node while it reads the residual
            identity = conv_bn_act(identity, self.downsample[0], self.downsample[1], relu=False, defer_apply=_DEFER_DS,
                                   compact_skip_grad=_COMPACT_DS)
        out = conv_bn_act(out, self.conv2, self.bn2, relu=True)
        return conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=identity, lazy_residual_grad=_LAZY_RES)


class ResNet(tnn.Module):
    def __init__(self, block, layers, num_classes: int = 1000, zero_init_residual: bool = False):
        super().__init__()
        self.inplanes = 64
        self.conv1 = pnn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = pnn.BatchNorm2d(64)
        self.relu = pnn.ReLU(inplace=True)
        self.maxpool = pnn.MaxPool3x3s2()
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = pnn.GlobalAvgPool()
        self.fc = pnn.Linear(512 * block.expansion, num_classes)

        for m in self.modules():
            if isinstance(m, tnn.Conv2d):
                tnn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, tnn.modules.batchnorm._BatchNorm):
                tnn.init.constant_(m.weight, 1)
                tnn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    tnn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    tnn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = tnn.Sequential(
                _conv1x1(self.inplanes, planes * block.expansion, stride),
                pnn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return tnn.Sequential(*layers)

    def forward(self, x):
        x = conv_bn_relu_maxpool(x, self.conv1, self.bn1, self.maxpool)  # BN + ReLU + pool: one fused node
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        x = self.avgpool(x)  # [B, C]
        return self.fc(x)


def resnet18(num_classes=1000, **kw):
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes, **kw)


def resnet34(num_classes=1000, **kw):
    return ResNet(BasicBlock, [3, 4, 6, 3], num_classes, **kw)


def resnet50(num_classes=1000, **kw):
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes, **kw)


def resnet101(num_classes=1000, **kw):
    return ResNet(Bottleneck, [3, 4, 23, 3], num_classes, **kw)


def resnet152(num_classes=1000, **kw):
    return ResNet(Bottleneck, [3, 8, 36, 3], num_classes, **kw)
