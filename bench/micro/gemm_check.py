"""Numerics + timing of the native bf16 GEMM (csrc/kernels/gemm.hip) against fp32 torch.matmul and the library GEMM.

    python bench/micro/gemm_check.py                 # ViT-B/16 shapes at B=128, ResNet fc, ragged cases
    python bench/micro/gemm_check.py --quick         # numerics only, small shapes

Every case checks ``epilogue(a @ b^T)`` with the error measured per element relative to the fp32 result's
row scale (so an error confined to a small-magnitude region is not hidden by a max-norm).
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.utils.env import require_native  # noqa: E402

_C = require_native()


def _time(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def check(M, N, K, bias=False, act=0, residual=False, gelu_in=False, preact=False, colsum=False, timing=True,
          tol=2e-2):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M * 31 + N * 7 + K)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev, generator=g) * (1.0 / K ** 0.5)).to(torch.bfloat16)
    bs = torch.randn(N, device=dev, generator=g) if bias else None
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if residual else None
    u = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if gelu_in else None
    c, pre, cs = _C.gemm_nt(a, b, bs, res, u, act, preact, colsum)
    ref = a.float() @ b.float().t()
    if bias:
        ref = ref + bs
    ref_pre = ref
    if act == 1:
        ref = torch.nn.functional.gelu(ref.to(torch.bfloat16).float())
    if gelu_in:
        uf = u.float().requires_grad_(True)
        torch.nn.functional.gelu(uf).sum().backward()
        ref = ref.to(torch.bfloat16).float() * uf.grad
    if residual:
        ref = ref.to(torch.bfloat16).float() + res.float()
    scale = ref.abs().mean(dim=1, keepdim=True).clamp_min(1e-6)
    err = ((c.float() - ref).abs() / (ref.abs() + scale)).max().item()
    ok = err < tol and bool(torch.isfinite(c.float()).all())
    msg = f"M={M} N={N} K={K} bias={int(bias)} act={act} res={int(residual)} dgelu={int(gelu_in)} err={err:.2e}"
    if preact:
        e2 = ((pre.float() - ref_pre).abs() / (ref_pre.abs() + ref_pre.abs().mean(dim=1, keepdim=True))).max().item()
        ok = ok and e2 < tol
        msg += f" preact_err={e2:.2e}"
    if colsum:
        want = c.float().sum(0)
        e3 = ((cs - want).abs() / (want.abs() + c.float().abs().sum(0) * 1e-3 + 1e-3)).max().item()
        ok = ok and e3 < 1e-2
        msg += f" colsum_err={e3:.2e}"
    if timing:
        t_nat = _time(lambda: _C.gemm_nt(a, b, bs, res, u, act, preact, colsum))
        t_lib = _time(lambda: torch.nn.functional.linear(a, b, None if bs is None else bs.to(torch.bfloat16)))
        fl = 2.0 * M * N * K
        msg += f" | native {t_nat:.1f} us ({fl / t_nat / 1e6:.0f} TF)  library {t_lib:.1f} us ({fl / t_lib / 1e6:.0f} TF)"
    print(("ok   " if ok else "FAIL ") + msg, flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--bn", type=int, default=0, help="force the tile width (128 / 256; 0: the launcher's choice)")
    a = ap.parse_args()
    _C.set_gemm_tile_width(a.bn)
    ok = True
    # ragged M / N / K, every epilogue
    for (M, N, K) in [(256, 256, 64), (300, 264, 128), (17, 8, 72), (513, 1000, 2048), (1000, 512, 1000)]:
        ok &= check(M, N, K, timing=False)
        ok &= check(M, N, K, bias=True, timing=False)
    ok &= check(777, 520, 192, bias=True, act=1, preact=True, timing=False)
    ok &= check(777, 520, 192, bias=True, residual=True, timing=False)
    ok &= check(777, 520, 192, gelu_in=True, colsum=True, timing=False)
    ok &= check(1280, 768, 320, colsum=True, timing=False)
    if not a.quick:
        M = a.batch * 197
        for (N, K, kw) in [(2304, 768, dict(bias=True)), (768, 768, dict(bias=True, residual=True)),
                           (3072, 768, dict(bias=True, act=1, preact=True)), (768, 3072, dict(bias=True, residual=True)),
                           (768, 2304, {}), (768, 768, {}), (3072, 768, dict(gelu_in=True, colsum=True)), (768, 3072, {})]:
            ok &= check(M, N, K, **kw)
        ok &= check(a.batch * 196, 768, 768, bias=True)
        ok &= check(1280, 1000, 2048, bias=True)
        ok &= check(8192, 8192, 8192)
        ok &= check(4096, 4096, 4096)
    print("ALL OK" if ok else "SOME FAILED", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
