"""3x3/2 max-pool backward at the stem shape: 2x2-blocked kernel vs the per-pixel one (PDT_POOL_BWD_2X2=0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pytorch_distributed_training_amd.ops import native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
x = torch.randn(B, 64, 112, 112, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
y, arg = native.maxpool3x3s2_fwd(x)
dy = torch.randn_like(y)
for _ in range(3):
    dx = native.maxpool3x3s2_bwd(dy, x, y, arg)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    dx = native.maxpool3x3s2_bwd(dy, x, y, arg)
b.record()
torch.cuda.synchronize()
xs = x[:4].float().requires_grad_(True)
torch.nn.functional.max_pool2d(xs, 3, 2, 1).backward(dy[:4].float())
print(f"B={B} maxpool bwd {a.elapsed_time(b) * 100:.1f} us (2x2={os.environ.get('PDT_POOL_BWD_2X2', '1')}), "
      f"max |err| vs torch on 4 images {float((dx[:4].float() - xs.grad).abs().max()):.4g}", flush=True)
