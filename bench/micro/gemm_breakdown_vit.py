"""Where the GEMM core spends a launch on ViT-B/16's linear shapes: the same launch with parts switched off
(PDT_GEMM_DEBUG bits: 1 stores dropped, 2 epilogue skipped, 4 MFMAs skipped, 8 operand DMA skipped).
Timing diagnostics only; the variants' results are wrong by construction."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.utils.env import require_native
_C = require_native()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = B * 197

def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

VARIANTS = [("full", 0), ("nostore", 1), ("noepi", 2), ("nomfma", 4), ("nodma", 8), ("nomfma+noepi", 6), ("nodma+noepi", 10), ("epi only", 12)]
print(f"M = {M} | " + " | ".join(n for n, _ in VARIANTS) + " | tiles | ideal us at 1.4 PF")
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    res = []
    for name, bits in VARIANTS:
        os.environ["PDT_GEMM_DEBUG"] = str(bits)
        res.append(t(lambda: _C.gemm_nt(a, b)))
    os.environ["PDT_GEMM_DEBUG"] = "0"
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"N={N:5d} K={K:5d} | " + " | ".join(f"{v:7.1f}" for v in res) + f" | {tiles} | {2.0 * M * N * K / 1.4e9:7.1f}", flush=True)
