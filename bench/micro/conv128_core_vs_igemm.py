"""128 -> 128 channel 3x3 / stride 1 at 28 x 28 (ResNet-50 layer 2): implicit-GEMM build the bindings pick against the
GEMM core's spatial-filter mode on its 256 x 128 tile build (cfg 100), forward with BN statistics and plain / fused data
gradient, at B = 1280."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kernel_check as kc
from pytorch_distributed_training_amd.ops import native

B, C, H = 1280, 128, 28
x = kc.nhwc(B, C, H, H)
dy = kc.nhwc(B, C, H, H)
add = kc.nhwc(B, C, H, H)
w = (torch.randn(C, C, 3, 3, device="cuda") * 0.03).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
xb = kc.nhwc(B, C, H, H)
st = native.bn_stats(xb)
mean, invstd = native.bn_finalize(st, float(B * H * H), 1e-5)
act = native.bn_apply_act(xb, mean, invstd, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), relu=True)
mask = act[1] if isinstance(act, tuple) else None
fl = 2.0 * B * H * H * C * C * 9
for cfg in (0, 100):
    t1 = kc.timeit(lambda: native.conv2d_fwd(x, w, 1, 1, True, cfg=cfg))
    t2 = kc.timeit(lambda: native.conv2d_dgrad(dy, w, (B, C, H, H), 1, 1, cfg=cfg))
    t3 = kc.timeit(lambda: native.conv2d_dgrad(dy, w, (B, C, H, H), 1, 1, add, cfg=cfg))
    print(f"cfg {cfg:3d}: fwd+stats {t1:7.1f} us ({fl / t1 / 1e6:5.0f} TF)  dgrad {t2:7.1f} us ({fl / t2 / 1e6:5.0f} TF)  dgrad+addend {t3:7.1f} us", flush=True)
