"""Per-layer roofline of the ResNet-50 training step on one MI355X.

Enumerates every distinct conv / BN shape of ResNet-50 (with its multiplicity), times the native
kernels on it and prints measured time against a bound of max(flops / MFMA_PEAK, bytes / HBM_BW),
sorted by the time that could still be won.  Output: a markdown table (stdout) + JSON under gpurun_out/.

    python bench/resnet50_roofline.py --batch 256
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MFMA_PEAK = 2.5e15   # dense bf16 FLOP/s
HBM_BW = 5.5e12      # B/s a streaming kernel reaches in practice (8 TB/s pin rate)


def resnet50_convs(res=224):
    """[(cin, hin, cout, k, stride, pad)] in execution order (stem excluded: Cin=3 goes elsewhere)."""
    out = []
    h = res // 4
    inpl = 64
    for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for b in range(blocks):
            s = stride if b == 0 else 1
            out.append((inpl, h, planes, 1, 1, 0))
            out.append((planes, h, planes, 3, s, 1))
            ho = h // s
            out.append((planes, ho, planes * 4, 1, 1, 0))
            if b == 0:
                out.append((inpl, h, planes * 4, 1, s, 0))
            inpl, h = planes * 4, ho
    return out


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3  # s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--out", default="gpurun_out/resnet50_roofline.json")
    a = ap.parse_args()
    from pytorch_distributed_training_amd.ops import native

    B = a.batch
    dev = "cuda"
    rows = []
    convs = collections.Counter(resnet50_convs())
    for (cin, h, cout, k, s, p), mult in convs.items():
        ho = (h + 2 * p - k) // s + 1
        x = torch.randn(B, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).to(torch.bfloat16).contiguous(
            memory_format=torch.channels_last)
        dy = torch.randn(B, cout, ho, ho, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        flops = 2.0 * B * ho * ho * cout * cin * k * k
        xin, yout, wb = x.numel() * 2, dy.numel() * 2, w.numel() * 2
        # stride-2 1x1 only touches a quarter of the input pixels, but whole cache lines of them
        cases = {
            "fwd": (lambda: native.conv2d_fwd(x, w, s, p, True), xin + yout + wb),
            "dgrad": (lambda: native.conv2d_dgrad(dy, w, (B, cin, h, h), s, p), xin + yout + wb),
            "wgrad": (lambda: native.conv2d_wgrad(dy, x, (cout, cin, k, k), s, p), xin + yout + wb * 2),
        }
        for name, (fn, nbytes) in cases.items():
            t = timeit(fn)
            bound = max(flops / MFMA_PEAK, nbytes / HBM_BW)
            rows.append({"op": f"conv_{name}", "shape": f"{cin}->{cout} {h}x{h} k{k} s{s}", "mult": mult,
                         "us": t * 1e6, "bound_us": bound * 1e6, "TF": flops / t / 1e12, "TBps": nbytes / t / 1e12,
                         "limit": "mfma" if flops / MFMA_PEAK > nbytes / HBM_BW else "hbm"})
        del x, w, dy
    # BN passes: one per conv output
    shapes = collections.Counter()
    for (cin, h, cout, k, s, p) in resnet50_convs():
        shapes[(cout, (h + 2 * p - k) // s + 1)] += 1
    for (c, h), mult in shapes.items():
        x = torch.randn(B, c, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res = torch.randn_like(x)
        dz = torch.randn_like(x)
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        cnt = float(B * h * h)
        mean, invstd = native.bn_finalize(native.bn_stats(x), cnt, 1e-5)
        z, act = native.bn_apply_act(x, mean, invstd, gamma, beta, None, True)
        ps = native.bn_bwd_reduce(dz, act, x, mean, invstd, True)
        n2 = x.numel() * 2
        cases = {
            "bn_apply": (lambda: native.bn_apply_act(x, mean, invstd, gamma, beta, None, True), n2 * 2 + n2 / 16),
            "bn_apply+res": (lambda: native.bn_apply_act(x, mean, invstd, gamma, beta, res, True), n2 * 3 + n2 / 16),
            "bn_bwd_reduce": (lambda: native.bn_bwd_reduce(dz, act, x, mean, invstd, True), n2 * 2 + n2 / 16),
            "bn_bwd_elemt": (lambda: native.bn_bwd_elemt(dz, act, x, mean, invstd, gamma, ps, cnt, True, False),
                             n2 * 3 + n2 / 16),
            "bn_bwd_elemt+dres": (lambda: native.bn_bwd_elemt(dz, act, x, mean, invstd, gamma, ps, cnt, True, True),
                                  n2 * 4 + n2 / 16),
        }
        for name, (fn, nbytes) in cases.items():
            t = timeit(fn)
            rows.append({"op": name, "shape": f"C{c} {h}x{h}", "mult": mult, "us": t * 1e6,
                         "bound_us": nbytes / HBM_BW * 1e6, "TF": 0.0, "TBps": nbytes / t / 1e12, "limit": "hbm"})
        del x, res, dz, z, act
    for r in rows:
        r["gap_us_total"] = (r["us"] - r["bound_us"]) * r["mult"]
    rows.sort(key=lambda r: -r["gap_us_total"])
    print(f"| op | shape (B={B}) | x | us | bound us | limit | TF | TB/s | winnable us (all layers) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['op']} | {r['shape']} | {r['mult']} | {r['us']:.1f} | {r['bound_us']:.1f} | {r['limit']} | "
              f"{r['TF']:.0f} | {r['TBps']:.2f} | {r['gap_us_total']:.0f} |", flush=True)
    tot = collections.defaultdict(float)
    bound = collections.defaultdict(float)
    for r in rows:
        key = r["op"].split("+")[0]
        tot[key] += r["us"] * r["mult"] if "+" not in r["op"] else 0.0
        bound[key] += r["bound_us"] * r["mult"] if "+" not in r["op"] else 0.0
    print("\ntotals per step (us; BN rows without the residual variants):")
    for k in tot:
        print(f"  {k:14s} measured {tot[k]:8.0f}   bound {bound[k]:8.0f}")
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
