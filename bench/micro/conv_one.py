"""Run a few ResNet-50 conv shapes (fwd / dgrad / wgrad) a handful of times: the target of rocprofv3 --pmc passes.

    python bench/micro/conv_one.py --batch 1024 --shapes 64,56,64,3,1 256,14,256,3,1 64,56,256,1,1
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--no_stats", action="store_true", help="forward without the fused BN-statistics epilogue")
    ap.add_argument("--time", action="store_true", help="print us per call of fwd / dgrad / wgrad (event-timed)")
    ap.add_argument("--shapes", nargs="+", default=["64,56,64,3,1", "256,14,256,3,1", "64,56,256,1,1", "256,56,64,1,1"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for spec in a.shapes:
        cin, h, cout, k, s = (int(v) for v in spec.split(","))
        pad = k // 2
        x = torch.randn(a.batch, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = native.conv2d_fwd(x, w, s, pad, True)[0]
        dy = torch.randn_like(y)
        for _ in range(a.iters):
            native.conv2d_fwd(x, w, s, pad, True)
            native.conv2d_dgrad(dy, w, tuple(x.shape), s, pad)
            native.conv2d_wgrad(dy, x, tuple(w.shape), s, pad)
        torch.cuda.synchronize()
        if a.time:
            res = []
            for fn in (lambda: native.conv2d_fwd(x, w, s, pad, not a.no_stats), lambda: native.conv2d_dgrad(dy, w, tuple(x.shape), s, pad),
                       lambda: native.conv2d_wgrad(dy, x, tuple(w.shape), s, pad)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn()
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 100.0)
            print(f"time {spec} cfg={os.environ.get('PDT_IGEMM_CFG', '0')} fwd {res[0]:.1f} dgrad {res[1]:.1f} wgrad {res[2]:.1f} us", flush=True)
        print("done", spec, flush=True)


if __name__ == "__main__":
    main()
