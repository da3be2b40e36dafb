"""Weight-gradient kernel on ResNet-50's layers at a given batch (and ViT-B/16's linear layers): us per launch and
TFLOP/s.  Run once per tile policy (the policy is read once per process): PDT_WGRAD_WIDE=0 = 128-wide tiles only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import native
from pytorch_distributed_training_amd.utils.env import require_native
_C = require_native()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024

def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

def nhwc(*s):
    return torch.randn(*s, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

total = 0.0
for (cin, hw, cout, k, s, n) in [(256, 14, 256, 3, 1, 5), (512, 7, 512, 3, 1, 2), (128, 28, 128, 3, 1, 3), (64, 56, 64, 3, 1, 3),
                                 (256, 14, 1024, 1, 1, 6), (1024, 14, 256, 1, 1, 5), (512, 7, 2048, 1, 1, 3), (2048, 7, 512, 1, 1, 2),
                                 (128, 28, 512, 1, 1, 4), (512, 28, 128, 1, 1, 3), (64, 56, 256, 1, 1, 4), (256, 56, 64, 1, 1, 2),
                                 (1024, 14, 512, 1, 1, 1), (512, 28, 256, 1, 1, 1), (256, 56, 128, 1, 1, 1),
                                 (256, 28, 256, 3, 2, 1), (512, 14, 512, 3, 2, 1), (128, 56, 128, 3, 2, 1),
                                 (256, 56, 512, 1, 2, 1), (512, 28, 1024, 1, 2, 1), (1024, 14, 2048, 1, 2, 1)]:
    p = 1 if k == 3 else 0
    x = nhwc(B, cin, hw, hw)
    ho = (hw + 2 * p - k) // s + 1
    dy = nhwc(B, cout, ho, ho)
    us = t(lambda: native.conv2d_wgrad(dy, x, (cout, cin, k, k), s, p))
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    total += us * n
    print(f"wgrad {cin:5d}->{cout:5d} @{hw:3d} k{k} s{s} x{n}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF", flush=True)
print(f"sum over ResNet-50 (weighted): {total / 1e3:.2f} ms")
for (m, kk, nn) in [(25216, 768, 2304), (25216, 768, 768), (25216, 768, 3072), (25216, 3072, 768)]:
    dy = torch.randn(m, nn, device="cuda").to(torch.bfloat16)
    x = torch.randn(m, kk, device="cuda").to(torch.bfloat16)
    us = t(lambda: _C.linear_wgrad(dy, x))
    print(f"linear_wgrad M={m} K={kk} N={nn}: {us:8.1f} us  {2.0 * m * kk * nn / us / 1e6:7.1f} TF", flush=True)
