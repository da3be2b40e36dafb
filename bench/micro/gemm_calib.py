"""Calibration only: what does the library GEMM (hipBLASLt through torch.matmul) reach on the pointwise-conv GEMM
shapes of ResNet-50 at B=1024?  Not used by the framework; a target for conv_igemm.hip."""
import torch

def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

B = 1024
for (hw, k, n) in ((56, 64, 256), (56, 256, 64), (28, 128, 512), (28, 512, 128), (14, 256, 1024), (14, 1024, 256),
                   (7, 512, 2048), (7, 2048, 512), (56, 64, 64)):
    m = B * hw * hw
    a = torch.randn(m, k, device="cuda").bfloat16()
    w = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: torch.matmul(a, w.t(), out=out))
    byts = (m * k + m * n + n * k) * 2
    print(f"M={m} K={k} N={n}: {us:8.1f} us  {2*m*k*n/us/1e6:6.0f} TF  {byts/us/1e6:5.2f} TB/s", flush=True)
