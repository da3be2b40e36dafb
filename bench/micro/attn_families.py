"""The three attention kernel families (whole-head pipelined, whole-head one head per workgroup, block streaming) on
inputs shaped like a ViT at initialisation (small q / k, near-uniform softmax, tiny output gradient) against fp32:
per-element errors relative to the largest element of the reference, forward and backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import native, reference
from pytorch_distributed_training_amd import _C

def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))

B, S, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 197, 12)))
for (qs, ds) in ((1.0, 1.0), (0.1, 1e-3), (0.02, 1e-4), (3.0, 1.0)):
    g = torch.Generator(device="cuda").manual_seed(7)
    qkv = (torch.randn(B, S, 3 * H * 64, device="cuda", generator=g) * qs).to(torch.bfloat16)
    do = (torch.randn(B, S, H * 64, device="cuda", generator=g) * ds).to(torch.bfloat16)
    outr, lser = reference.attention_fwd(qkv.float(), H)
    dr = reference.attention_bwd(do.float(), qkv.float(), outr, lser, H)
    modes = [("generic", None)] if os.environ.get("PDT_ATTN_GENERIC") == "1" else [("pipelined", True), ("one-head", False)]
    for name, flag in modes:
        if flag is not None:
            _C.set_attention_pipelined(flag)
        out, lse = native.attention_fwd(qkv, H)
        d = native.attention_bwd(do, qkv, out, lse, H)
        D = H * 64
        print(f"q scale {qs:5.2f} dout scale {ds:7.0e} {name:10s}: out {rel(out, outr):.2e} lse {rel(lse, lser):.2e} "
              f"dq {rel(d[..., :D], dr[..., :D]):.2e} dk {rel(d[..., D:2*D], dr[..., D:2*D]):.2e} dv {rel(d[..., 2*D:], dr[..., 2*D:]):.2e} "
              f"finite {bool(torch.isfinite(d.float()).all())}", flush=True)
