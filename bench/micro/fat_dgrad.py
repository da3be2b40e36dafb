"""Data gradient of the bottleneck's first 1x1 convolution with everything its epilogue fuses (residual-branch gradient
gated by the block's ReLU mask + BN-backward sums of the previous block's bn3): GEMM core vs implicit-GEMM builds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280

def t(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

def nhwc(*s):
    return torch.randn(*s, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

for (c4, hw, c) in [(256, 56, 64), (512, 28, 128), (1024, 14, 256), (2048, 7, 512)]:
    w = (torch.randn(c, c4, 1, 1, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = nhwc(B, c, hw, hw)
    xb = nhwc(B, c4, hw, hw)
    add = nhwc(B, c4, hw, hw)
    mean, invstd = native.bn_finalize(native.bn_stats(xb), float(B * hw * hw), 1e-5)
    one = torch.ones(c4, device="cuda")
    _, act = native.bn_apply_act(xb, mean, invstd, one, one * 0.1, None, True)
    M = B * hw * hw
    byts = M * (c * 2 + c4 * 2 * 3 + c4 // 8 * 2)
    res = {}
    for name, cfg in (("core", 100), ("igemm<128,128,4,2>", 11), ("igemm<256,128,2,3>", 10), ("igemm auto", 0)):
        if cfg == 0:
            os.environ["X"] = "1"
        try:
            res[name] = t(lambda: native.conv2d_dgrad(dy, w, (B, c4, hw, hw), 1, 0, add, (xb, act, mean, invstd), act, cfg=cfg))
        except Exception as e:
            res[name] = float("nan")
    print(f"dgrad {c:4d}->{c4:5d} @{hw:3d} fused: " + "  ".join(f"{k} {v:7.1f} us" for k, v in res.items()) + f"   stream bound {byts / 5.5e6:7.1f} us", flush=True)
