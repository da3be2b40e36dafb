"""What does the memory system give a streaming kernel on this GPU?  copy (1R+1W), add (2R+1W), and the BN kernels
on the largest ResNet-50 activation, in TB/s: the ceiling the HBM-bound kernels are judged against."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import native  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


for B in (256, 1024):
    x = torch.randn(B, 256, 56, 56, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x)
    r = torch.randn_like(x)
    n = x.numel() * 2
    t = timeit(lambda: y.copy_(x))
    print(f"B={B} copy   {2 * n / t / 1e12:.2f} TB/s")
    t = timeit(lambda: y.fill_(1.0))
    print(f"B={B} fill   {n / t / 1e12:.2f} TB/s (write only)")
    t = timeit(lambda: torch.add(x, r, out=y))
    print(f"B={B} add    {3 * n / t / 1e12:.2f} TB/s")
    t = timeit(lambda: x.sum())
    print(f"B={B} read   {n / t / 1e12:.2f} TB/s (torch sum)")
    c = 256
    gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    mean, invstd = native.bn_finalize(native.bn_stats(x), float(B * 56 * 56), 1e-5)
    t = timeit(lambda: native.bn_apply_act(x, mean, invstd, gamma, beta, r, True))
    print(f"B={B} bn_apply+res  {(3 + 1 / 16) * n / t / 1e12:.2f} TB/s")
    z, act = native.bn_apply_act(x, mean, invstd, gamma, beta, r, True)
    ps = native.bn_bwd_reduce(r, act, x, mean, invstd, True)
    t = timeit(lambda: native.bn_bwd_elemt(r, act, x, mean, invstd, gamma, ps, float(B * 3136), True, True))
    print(f"B={B} bn_bwd_elemt+dres  {(4 + 1 / 16) * n / t / 1e12:.2f} TB/s")
    t = timeit(lambda: native.bn_bwd_reduce(r, act, x, mean, invstd, True))
    print(f"B={B} bn_bwd_reduce  {(2 + 1 / 16) * n / t / 1e12:.2f} TB/s")
    del x, y, r, z, act

# where does bn_apply lose against a plain add?  (mask write on/off, residual on/off)
B = 1024
x = torch.randn(B, 256, 56, 56, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
r = torch.randn_like(x)
n = x.numel() * 2
gamma, beta = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
mean, invstd = native.bn_finalize(native.bn_stats(x), float(B * 56 * 56), 1e-5)
for relu in (True, False):
    for res in (r, None):
        t = timeit(lambda: native.bn_apply_act(x, mean, invstd, gamma, beta, res, relu))
        units = 2 + (1 if res is not None else 0) + (1 / 16 if relu else 0)
        print(f"B={B} bn_apply relu={relu} res={res is not None}: {units * n / t / 1e12:.2f} TB/s  {t * 1e6:.0f} us")
