"""Time the GEMM core (csrc/kernels/gemm.hip) on the GEMM view of ResNet-50's pointwise convolutions at a given
batch, next to the implicit-GEMM kernel (conv_igemm.hip) on the same layer: decides which layers move over."""
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

for (cin, hw, cout) in [(64, 56, 256), (256, 56, 64), (128, 28, 512), (512, 28, 128), (256, 14, 1024), (1024, 14, 256),
                        (512, 7, 2048), (2048, 7, 512), (1024, 14, 512), (512, 28, 256), (256, 56, 128)]:
    x = torch.randn(B, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    M = B * hw * hw
    a = x.permute(0, 2, 3, 1).reshape(M, cin)
    w2 = w.reshape(cout, cin)
    us_ig = t(lambda: native.conv2d_fwd(x, w, 1, 0, True))
    us_ig0 = t(lambda: native.conv2d_fwd(x, w, 1, 0, False))
    us_g = t(lambda: _C.gemm_nt(a, w2))
    us_gc = t(lambda: _C.gemm_nt(a, w2, None, None, None, 0, False, True))
    y = native.conv2d_fwd(x, w, 1, 0, False)[0].permute(0, 2, 3, 1).reshape(M, cout)
    err = float((y.float() - _C.gemm_nt(a, w2)[0].float()).abs().max())
    byts = (M * cin + M * cout) * 2
    print(f"{cin:5d}->{cout:5d} @{hw:3d}  igemm+stats {us_ig:7.1f}  igemm {us_ig0:7.1f}  gemm {us_g:7.1f}  gemm+colsum {us_gc:7.1f} us   hbm bound {byts/5.5e6:7.1f} us  maxdiff {err:.3g}", flush=True)

# 3x3 / stride 1 / pad 1: the core's spatial-filter mode (cfg 100) against the implicit-GEMM kernel (cfg 10 = its
# 256x128 build, what the shape heuristic picks for these layers)
for (c, hw) in [(256, 14), (512, 7), (128, 28), (64, 56)]:
    x = torch.randn(B, c, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    fl = 2.0 * B * hw * hw * c * c * 9
    r = {}
    for name, cfg in (("igemm", 15 if c == 64 else (12 if c < 256 else 10)), ("core", 100)) + ((("win", 101),) if c == 64 else ()):
        r[name + " fwd"] = t(lambda: native.conv2d_fwd(x, w, 1, 1, True, cfg=cfg))
        r[name + " dgrad"] = t(lambda: native.conv2d_dgrad(dy, w, (B, c, hw, hw), 1, 1, cfg=cfg))
    err = float((native.conv2d_fwd(x, w, 1, 1, False, cfg=101 if c == 64 else 100)[0].float() - native.conv2d_fwd(x, w, 1, 1, False, cfg=15 if c == 64 else (12 if c < 256 else 10))[0].float()).abs().max())
    print(f"3x3 {c:4d}ch @{hw:3d}: " + "  ".join(f"{k} {v:7.1f} us ({fl / v / 1e6:5.0f} TF)" for k, v in r.items()) + f"  maxdiff {err:.3g}", flush=True)
