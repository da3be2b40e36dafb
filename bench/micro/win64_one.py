import os, sys, torch
sys.path.insert(0, "/root/repo")
from pytorch_distributed_training_amd.ops import native
B = 1280
x = torch.randn(B, 64, 56, 56, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for r in range(2):
    print("win64 fwd+stats us", round(t(lambda: native.conv2d_fwd(x, w, 1, 1, True)), 1), "dbg", os.environ.get("PDT_WIN64_DBG"), flush=True)
