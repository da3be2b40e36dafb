"""One weight-gradient shape, a few launches (for rocprofv3 --pmc passes): wgrad_one.py B cin hw cout k stride"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import native
B, cin, hw, cout, k, s = [int(v) for v in sys.argv[1:7]]
p = 1 if k == 3 else 0
x = torch.randn(B, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
ho = (hw + 2 * p - k) // s + 1
dy = torch.randn(B, cout, ho, ho, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for _ in range(5):
    native.conv2d_wgrad(dy, x, (cout, cin, k, k), s, p)
torch.cuda.synchronize()
