"""Where a GEMM-core launch spends its time on the pointwise / 3x3 convolution shapes of ResNet-50: the same launch
with parts switched off (PDT_GEMM_DEBUG bits: 1 stores dropped, 2 epilogue skipped, 4 MFMAs skipped, 8 operand DMA
skipped).  Timing diagnostics only; the variants' results are wrong by construction."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import native
from pytorch_distributed_training_amd.utils.env import require_native
_C = require_native()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280

def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

VARIANTS = [("full", 0), ("nostore", 1), ("noepi", 2), ("nomfma", 4), ("nodma", 8), ("nomfma+noepi", 6), ("nodma+noepi", 10),
            ("nodma+nomfma", 12), ("epi only", 12), ("loads only", 6)]
print("shape | " + " | ".join(n for n, _ in VARIANTS[:8]) + " | hbm bound | mfma@1.45PF")
for (cin, hw, cout) in [(64, 56, 256), (128, 28, 512), (256, 14, 1024), (1024, 14, 256), (512, 7, 2048), (2048, 7, 512),
                        (1024, 14, 512), (512, 28, 256)]:
    x = torch.randn(B, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    M = B * hw * hw
    res = []
    for name, bits in VARIANTS[:8]:
        os.environ["PDT_GEMM_DEBUG"] = str(bits)
        res.append(t(lambda: native.conv2d_fwd(x, w, 1, 0, True)))
    os.environ["PDT_GEMM_DEBUG"] = "0"
    byts = (M * cin + M * cout) * 2
    fl = 2.0 * M * cin * cout
    print(f"1x1 fwd+stats {cin:5d}->{cout:5d} @{hw:3d} | " + " | ".join(f"{v:7.1f}" for v in res) + f" | {byts / 5.5e6:7.1f} | {fl / 1.45e9:7.1f}", flush=True)
for (c, hw) in [(256, 14), (512, 7)]:
    x = torch.randn(B, c, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device="cuda") * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    M = B * hw * hw
    res = []
    for name, bits in VARIANTS[:8]:
        os.environ["PDT_GEMM_DEBUG"] = str(bits)
        res.append(t(lambda: native.conv2d_fwd(x, w, 1, 1, True, cfg=100)))
    os.environ["PDT_GEMM_DEBUG"] = "0"
    fl = 2.0 * M * c * c * 9
    print(f"3x3 fwd+stats {c:5d} @{hw:3d} | " + " | ".join(f"{v:7.1f}" for v in res) + f" | {4.0 * M * c / 5.5e6:7.1f} | {fl / 1.45e9:7.1f}", flush=True)
