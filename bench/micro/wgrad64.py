"""64 -> 64 channel 3x3 weight gradient at 56x56: window kernel (wgrad_win64.hip) vs the per-tap split-K kernel.
usage: python bench/micro/wgrad64.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pytorch_distributed_training_amd import _C  # noqa: E402
from pytorch_distributed_training_amd.ops import native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
mk = lambda: torch.randn(B, 64, 56, 56, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
x, dy = mk(), mk()


def t(mode, n=10):
    _C.set_wgrad_win64_mode(mode)
    for _ in range(3):
        native.conv2d_wgrad(dy, x, (64, 64, 3, 3), 1, 1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        native.conv2d_wgrad(dy, x, (64, 64, 3, 3), 1, 1)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


flops = 2.0 * B * 56 * 56 * 64 * 64 * 9
for rep in range(2):
    for mode, name in ((0, "per-tap"), (2, "window")):
        us = t(mode)
        print(f"B={B} {name}: {us:.1f} us ({flops / us / 1e6:.0f} TFLOP/s, incl. slab reduction)", flush=True)
