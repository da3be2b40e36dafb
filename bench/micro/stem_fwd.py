"""Stem forward (7x7/2 on 3 channels -> 64): window kernel (conv_stem_win.hip) vs the implicit-GEMM launch on the same
packed operands.  usage: python bench/micro/stem_fwd.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from pytorch_distributed_training_amd import _C
from pytorch_distributed_training_amd.ops import native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
x = torch.randn(B, 3, 224, 224, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 3, 7, 7, device="cuda") / 12).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def t(mode, n=10):
    _C.set_stem_win_mode(mode)
    for _ in range(3):
        native.conv2d_fwd(x, w, 2, 3, True, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        native.conv2d_fwd(x, w, 2, 3, True, True)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for rep in range(2):
    for mode, name in ((0, "igemm"), (2, "window")):
        print(f"B={B} {name}: {t(mode):.1f} us (pack + conv + stats)", flush=True)


dy = torch.randn(B, 64, 112, 112, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
xp = native.conv2d_fwd(x, w, 2, 3, False, True)[2]


def tw(mode, n=10):
    _C.set_stem_win_mode(mode)
    for _ in range(3):
        native.conv2d_wgrad(dy, xp, (64, 3, 7, 7), 2, 3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        native.conv2d_wgrad(dy, xp, (64, 3, 7, 7), 2, 3)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for rep in range(2):
    for mode, name in ((0, "wgrad_win mode 1"), (2, "stem window")):
        print(f"B={B} weight gradient, {name}: {tw(mode):.1f} us", flush=True)
