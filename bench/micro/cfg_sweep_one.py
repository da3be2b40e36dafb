"""implicit-GEMM build sweep on one layer: cfg_sweep_one.py B cin hw cout k stride"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import native
B, cin, hw, cout, k, s = [int(v) for v in sys.argv[1:7]]
p = 1 if k == 3 else 0
def t(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
x = torch.randn(B, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
ho = (hw + 2 * p - k) // s + 1
dy = torch.randn(B, cout, ho, ho, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
cfgs = [int(v) for v in sys.argv[7].split(',')] if len(sys.argv) > 7 else (0, 5, 6, 8, 9, 10, 11, 12, 15)
t(lambda: native.conv2d_fwd(x, w, s, p, True, cfg=0))  # (the first timed call of a process reads 5-10 % slow)
for cfg in cfgs:
    try:
        f = t(lambda: native.conv2d_fwd(x, w, s, p, True, cfg=cfg))
        d = t(lambda: native.conv2d_dgrad(dy, w, (B, cin, hw, hw), s, p, cfg=cfg))
        print(f"cfg {cfg:3d}: fwd {f:8.1f} us  dgrad {d:8.1f} us", flush=True)
    except Exception as e:
        print(f"cfg {cfg:3d}: {str(e)[:80]}", flush=True)
