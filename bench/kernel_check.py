"""GPU numerics + timing sweep of every native kernel against the PyTorch oracles.

Runs every case even when an earlier one fails (a gpurun round trip is
expensive), prints one line per case and a JSON summary under gpurun_out/.

    python bench/kernel_check.py [--quick] [--time] [--only conv,bn,...]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pytorch_distributed_training_amd.ops import native, reference  # noqa: E402

DEV = "cuda"
RESULTS = []


def rel_err(a, b):
    a, b = a.float(), b.float()
    denom = b.abs().max().clamp_min(1e-6)
    return float((a - b).abs().max() / denom)


def elem_err(a, b):
    """max over elements of |a - b| / (|b| + mean|b|): a per-element relative error (an error confined to a
    small-magnitude region -- borders, padded rows, a ragged tile -- is not hidden behind the tensor's maximum)"""
    a, b = a.float(), b.float()
    return float(((a - b).abs() / (b.abs() + b.abs().mean().clamp_min(1e-12))).max())


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def record(name, ok, err, extra=None):
    r = {"case": name, "ok": bool(ok), "err": err}
    if extra:
        r.update(extra)
    RESULTS.append(r)
    print(("PASS " if ok else "FAIL ") + name + f"  err={err:.3e}" + (f"  {extra}" if extra else ""), flush=True)


def run_case(name, fn):
    try:
        fn()
    except Exception as e:  # keep going: one call must report everything
        traceback.print_exc()
        record(name, False, float("nan"), {"exception": repr(e)[:300]})
    torch.cuda.synchronize()


def nhwc(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------------------ conv
def conv_cases(quick):
    cases = [
        # B, Cin, H, W, Cout, k, stride, pad
        (2, 64, 8, 8, 64, 1, 1, 0),
        (2, 64, 9, 7, 128, 3, 1, 1),
        (3, 128, 10, 10, 64, 3, 2, 1),
        (2, 256, 8, 8, 512, 1, 2, 0),
        (2, 64, 13, 13, 64, 3, 1, 1),
        (4, 128, 14, 14, 256, 1, 1, 0),
        (2, 512, 7, 7, 512, 3, 1, 1),
    ]
    if not quick:
        cases += [
            (128, 64, 56, 56, 64, 3, 1, 1),
            (128, 64, 56, 56, 256, 1, 1, 0),
            (128, 256, 56, 56, 64, 1, 1, 0),
            (128, 128, 28, 28, 128, 3, 1, 1),
            (128, 512, 28, 28, 128, 1, 1, 0),
            (128, 128, 56, 56, 128, 3, 2, 1),
            (128, 256, 14, 14, 256, 3, 1, 1),
            (128, 256, 14, 14, 1024, 1, 1, 0),
            (128, 1024, 14, 14, 2048, 1, 2, 0),
            (128, 512, 7, 7, 512, 3, 1, 1),
            (128, 512, 7, 7, 2048, 1, 1, 0),
        ]
    return cases


GEMM_CORE = 100  # cfg: the GEMM core (csrc/kernels/gemm.hip) instead of an implicit-GEMM build
WIN64 = 101      # cfg: the 64-channel 3x3 window kernel (csrc/kernels/conv_win64.hip)


def check_conv(args, time_it, cfg=0, splits=0):
    """``cfg``: force one implicit-GEMM build (csrc/kernels/kernels.h), ``splits``: force the weight gradient's split-K"""
    B, C, H, W, N, k, s, p = args
    tag = f"B{B} {C}->{N} {H}x{W} k{k} s{s} p{p}" + (f" cfg{cfg}" if cfg else "") + (f" splits{splits}" if splits else "")
    x = nhwc(B, C, H, W)
    w = (torch.randn(N, C, k, k, device=DEV) * (1.0 / (C * k * k) ** 0.5)).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)

    def fwd():
        y, st = native.conv2d_fwd(x, w, s, p, True, cfg=cfg)
        yr, _ = reference.conv2d_fwd(x.float(), w.float(), s, p, False)
        e = max(rel_err(y, yr), elem_err(y, yr))
        str_ = reference.bn_stats(y)[0]
        es = rel_err(reference.bn_sum_partials(st), str_)
        extra = {"stats_err": es}
        if time_it:
            us = timeit(lambda: native.conv2d_fwd(x, w, s, p, True))
            us_lib = timeit(lambda: torch.nn.functional.conv2d(x, w, None, s, p))
            fl = 2.0 * y.numel() * C * k * k
            extra.update({"us": round(us, 1), "us_miopen": round(us_lib, 1), "TF": round(fl / us / 1e6, 1)})
        record(f"conv_fwd   {tag}", e < 2e-2 and es < 2e-2, e, extra)

    def dgrad():
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = nhwc(B, N, Ho, Wo)
        dx = native.conv2d_dgrad(dy, w, (B, C, H, W), s, p, cfg=cfg)
        dxr = reference.conv2d_dgrad(dy.float(), w.float(), (B, C, H, W), s, p)
        e = max(rel_err(dx, dxr), elem_err(dx, dxr))
        extra = {}
        if time_it:
            us = timeit(lambda: native.conv2d_dgrad(dy, w, (B, C, H, W), s, p))
            us_lib = timeit(lambda: reference.conv2d_dgrad(dy, w, (B, C, H, W), s, p))
            fl = 2.0 * dy.numel() * C * k * k
            extra.update({"us": round(us, 1), "us_miopen": round(us_lib, 1), "TF": round(fl / us / 1e6, 1)})
        record(f"conv_dgrad {tag}", e < 2e-2, e, extra)

    def wgrad():
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = nhwc(B, N, Ho, Wo)
        dw = native.conv2d_wgrad(dy, x, (N, C, k, k), s, p, splits=splits)
        dwr = reference.conv2d_wgrad(dy.float(), x.float(), (N, C, k, k), s, p)
        e = max(rel_err(dw, dwr), elem_err(dw, dwr) * 0.5)  # (sums of ~M products: many exact results are near zero)
        extra = {}
        if time_it:
            us = timeit(lambda: native.conv2d_wgrad(dy, x, (N, C, k, k), s, p))
            us_lib = timeit(lambda: reference.conv2d_wgrad(dy, x, (N, C, k, k), s, p))
            fl = 2.0 * dy.numel() * C * k * k
            extra.update({"us": round(us, 1), "us_miopen": round(us_lib, 1), "TF": round(fl / us / 1e6, 1)})
        record(f"conv_wgrad {tag}", e < 2e-2, e, extra)

    def dgrad_bnr():
        """stride-1 dgrad with the BN-backward reduction of the producing BN+ReLU fused into the epilogue"""
        if s != 1:
            return
        dy = nhwc(B, N, H, W) if k == 1 or p * 2 + 1 == k else None
        if dy is None:
            return
        xb = nhwc(B, C, H, W) * 2 + 0.3          # the BN's input
        xb = xb.contiguous(memory_format=torch.channels_last)
        gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
        mean, invstd = reference.bn_finalize(reference.bn_stats(xb), float(B * H * W), 1e-5)
        z, act = native.bn_apply_act(xb, mean, invstd, gamma, beta, None, True)
        add = nhwc(B, C, H, W)
        dx, ps = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, add, (xb, act, mean, invstd), cfg=cfg)
        dx2 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, add, cfg=cfg)
        ps2 = native.bn_bwd_reduce(dx2, act, xb, mean, invstd, True)
        e = max(rel_err(dx, dx2), rel_err(native.bn_sum_partials(ps), native.bn_sum_partials(ps2)))
        extra = {}
        if time_it:
            extra["us_fused"] = round(timeit(lambda: native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, add, (xb, act, mean, invstd))), 1)
            extra["us_dgrad"] = round(timeit(lambda: native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, add)), 1)
            extra["us_reduce"] = round(timeit(lambda: native.bn_bwd_reduce(dx2, act, xb, mean, invstd, True)), 1)
        record(f"conv_dgrad+bn_reduce {tag}", e < 2e-3, e, extra)
        # gated addend (lazily masked residual gradient): same as adding the materialised masked tensor
        gate_src = nhwc(B, C, H, W)
        _, gmask = native.bn_apply_act(gate_src, mean * 0, invstd * 0 + 1, gamma * 0 + 1, beta * 0, None, True)
        dx3 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, add, addend_mask=gmask, cfg=cfg)
        dx4 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, native.apply_relu_gate(add, gmask), cfg=cfg)
        dx5, ps5 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, add, (xb, act, mean, invstd), gmask, cfg=cfg)
        ps6 = native.bn_bwd_reduce(dx4, act, xb, mean, invstd, True)
        e2 = max(rel_err(dx3, dx4), rel_err(dx5, dx4),
                 rel_err(native.bn_sum_partials(ps5), native.bn_sum_partials(ps6)))
        ref_gate = (gate_src > 0)
        e2 = max(e2, rel_err(native.apply_relu_gate(add, gmask), add * ref_gate))
        if cfg in (GEMM_CORE, WIN64):  # the same results from the implicit-GEMM kernel (bf16 roundings of identical sums)
            dx0, ps0 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, add, (xb, act, mean, invstd), gmask, cfg=6)
            e2 = max(e2, rel_err(dx5, dx0), rel_err(native.bn_sum_partials(ps5), native.bn_sum_partials(ps0)))
            dxb, psb = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, None, (xb, act, mean, invstd), cfg=cfg)
            dxc, psc = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, None, (xb, act, mean, invstd), cfg=6)
            e2 = max(e2, rel_err(dxb, dxc), rel_err(native.bn_sum_partials(psb), native.bn_sum_partials(psc)))
        record(f"conv_dgrad+gated_addend {tag}", e2 < 2e-3, e2, {})
        # compact stride-2 addend (gradient of a 1x1 stride-2 conv of the same input) == its zero-stuffed form
        if H % 2 == 0 and W % 2 == 0:
            comp = nhwc(B, C, H // 2, W // 2)
            full = torch.zeros_like(add)
            full[:, :, ::2, ::2] = comp
            dx7, ps7 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, comp, (xb, act, mean, invstd), None, 2)
            dx8, ps8 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, full, (xb, act, mean, invstd))
            dx9 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, comp, addend_stride=2)
            e3 = max(rel_err(dx7, dx8), rel_err(dx9, dx8),
                     rel_err(native.bn_sum_partials(ps7), native.bn_sum_partials(ps8)))
            record(f"conv_dgrad+compact_addend {tag}", e3 < 1e-6, e3, {})

    run_case(f"conv_fwd   {tag}", fwd)
    run_case(f"conv_dgrad {tag}", dgrad)
    run_case(f"conv_dgrad+bnr {tag}", dgrad_bnr)
    run_case(f"conv_wgrad {tag}", wgrad)


def check_stem(B, H, W, time_it):
    """7x7 / 2 / pad 3 on 3 channels: packed-image forward (+BN statistics) and window weight gradient."""
    tag = f"B{B} 3->64 {H}x{W} k7 s2 p3"
    x = nhwc(B, 3, H, W)
    w = (torch.randn(64, 3, 7, 7, device=DEV) * (1.0 / 147 ** 0.5)).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    dy = nhwc(B, 64, H // 2, W // 2)

    def fwd():
        assert native.conv_unsupported_reason(x, w, 2, 3) is None
        y, st, xp = native.conv2d_fwd(x, w, 2, 3, True, True)
        yr, _ = reference.conv2d_fwd(x.float(), w.float(), 2, 3, False)
        e = rel_err(y, yr)
        es = rel_err(reference.bn_sum_partials(st), reference.bn_stats(y)[0])
        extra = {"stats_err": es}
        if time_it:
            us = timeit(lambda: native.conv2d_fwd(x, w, 2, 3, True))
            us_lib = timeit(lambda: torch.nn.functional.conv2d(x, w, None, 2, 3))
            extra.update({"us": round(us, 1), "us_miopen": round(us_lib, 1)})
        record(f"stem_fwd   {tag}", e < 2e-2 and es < 2e-2, e, extra)

    def wgrad():
        xp = native.conv2d_fwd(x, w, 2, 3, False, True)[2]
        dw = native.conv2d_wgrad(dy, xp, (64, 3, 7, 7), 2, 3)
        dw2 = native.conv2d_wgrad(dy, x, (64, 3, 7, 7), 2, 3)  # packs by itself
        dwr = reference.conv2d_wgrad(dy.float(), x.float(), (64, 3, 7, 7), 2, 3)
        e = max(rel_err(dw, dwr), rel_err(dw2, dwr))
        extra = {}
        if time_it:
            us = timeit(lambda: native.conv2d_wgrad(dy, xp, (64, 3, 7, 7), 2, 3))
            us_lib = timeit(lambda: reference.conv2d_wgrad(dy, x, (64, 3, 7, 7), 2, 3))
            extra.update({"us": round(us, 1), "us_miopen": round(us_lib, 1)})
        record(f"stem_wgrad {tag}", e < 2e-2, e, extra)

    run_case(f"stem_fwd {tag}", fwd)
    run_case(f"stem_wgrad {tag}", wgrad)


# -------------------------------------------------------------------------------- bn
def check_bn(B, C, H, W, time_it):
    tag = f"B{B} C{C} {H}x{W}"
    x = nhwc(B, C, H, W) * 2 + 0.5
    x = x.contiguous(memory_format=torch.channels_last)
    res = nhwc(B, C, H, W)
    gamma = torch.rand(C, device=DEV) + 0.5
    beta = torch.randn(C, device=DEV) * 0.1
    count = float(B * H * W)

    def stats():
        st = native.bn_sum_partials(native.bn_stats(x))
        e = rel_err(st, reference.bn_stats(x)[0])
        extra = {"us": round(timeit(lambda: native.bn_stats(x)), 1)} if time_it else None
        record(f"bn_stats   {tag}", e < 1e-3, e, extra)

    def fin_apply():
        st = reference.bn_stats(x)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        rm2, rv2 = rm.clone(), rv.clone()
        nbt, nbt2 = torch.zeros((), dtype=torch.int64, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
        mean, invstd = native.bn_finalize(native.bn_stats(x), count, 1e-5, rm, rv, 0.1, nbt)
        mr, ir = reference.bn_finalize(st, count, 1e-5, rm2, rv2, 0.1, nbt2)
        e = max(rel_err(mean, mr), rel_err(invstd, ir), rel_err(rm, rm2), rel_err(rv, rv2), float(abs(int(nbt) - int(nbt2))))
        record(f"bn_finalize {tag}", e < 1e-4, e)
        for relu, use_res in ((True, True), (True, False), (False, False), (False, True)):
            z, act = native.bn_apply_act(x, mr, ir, gamma, beta, res if use_res else None, relu)
            zr, _ = reference.bn_apply_act(x, mr, ir, gamma, beta, res if use_res else None, relu)
            e = rel_err(z, zr)
            if relu:  # the bitmask must agree with the z this kernel wrote
                bits = ((act.view(-1, C // 8, 1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).reshape(-1, C)
                zf = z.permute(0, 2, 3, 1).reshape(-1, C)
                e = max(e, float(((zf > 0) != (bits > 0)).sum()))
            extra = None
            if time_it and relu and use_res:
                us = timeit(lambda: native.bn_apply_act(x, mr, ir, gamma, beta, res, True))
                extra = {"us": round(us, 1), "GBps": round(3 * x.numel() * 2 / us / 1e3, 1)}
            record(f"bn_apply r{int(relu)} res{int(use_res)} {tag}", e < 1e-2, e, extra)
        # residual that is the un-normalised input of another BN (deferred downsample BN)
        ra = (torch.randn(C, device=DEV) * 0.2, torch.rand(C, device=DEV) + 0.5, torch.rand(C, device=DEV) + 0.5,
              torch.randn(C, device=DEV) * 0.1)
        z, _ = native.bn_apply_act(x, mr, ir, gamma, beta, res, True, ra)
        zr, _ = reference.bn_apply_act(x, mr, ir, gamma, beta, res, True, ra)
        record(f"bn_apply+res_affine {tag}", rel_err(z, zr) < 1e-2, rel_err(z, zr))

    def bwd():
        st = reference.bn_stats(x)
        mean, invstd = reference.bn_finalize(st, count, 1e-5)
        z, act = native.bn_apply_act(x, mean, invstd, gamma, beta, res, True)
        dz = nhwc(B, C, H, W)
        for relu in (True, False):
            ps = native.bn_bwd_reduce(dz, act, x, mean, invstd, relu)
            sr = reference.bn_bwd_reduce(dz, z, x, mean, invstd, relu)
            e = rel_err(native.bn_sum_partials(ps), sr[0])
            extra = {"us": round(timeit(lambda: native.bn_bwd_reduce(dz, act, x, mean, invstd, relu)), 1)} if time_it else None
            record(f"bn_bwd_reduce r{int(relu)} {tag}", e < 2e-3, e, extra)
            for dres in (True, False):
                dx, dr, sm = native.bn_bwd_elemt(dz, act, x, mean, invstd, gamma, ps, count, relu, dres)
                dxr, drr, smr = reference.bn_bwd_elemt(dz, z, x, mean, invstd, gamma, sr, count, relu, dres)
                e = max(rel_err(dx, dxr), rel_err(sm, smr))
                if dres:
                    e = max(e, rel_err(dr, drr))
                record(f"bn_bwd_elemt r{int(relu)} dres{int(dres)} {tag}", e < 1e-2, e)

    run_case(f"bn_stats {tag}", stats)
    run_case(f"bn_finalize/apply {tag}", fin_apply)
    run_case(f"bn_bwd {tag}", bwd)


def check_attention(B, S, H, time_it, hd=64):
    tag = f"B{B} S{S} H{H}" + ("" if hd == 64 else f" hd{hd}")
    qkv = torch.randn(B, S, 3 * H * hd, device=DEV).to(torch.bfloat16)
    dout = torch.randn(B, S, H * hd, device=DEV).to(torch.bfloat16)

    def run():
        out, lse = native.attention_fwd(qkv, H)
        outr, lser = reference.attention_fwd(qkv.float(), H)
        e = max(rel_err(out, outr), rel_err(lse, lser))
        extra = None
        if time_it:
            extra = {"us": round(timeit(lambda: native.attention_fwd(qkv, H)), 1)}
        record(f"attention_fwd {tag}", e < 2e-2, e, extra)
        dqkv = native.attention_bwd(dout, qkv, out, lse, H)
        dr = reference.attention_bwd(dout.float(), qkv.float(), outr, lser, H)
        d = 3 * H * hd
        parts = {"dq": slice(0, d // 3), "dk": slice(d // 3, 2 * d // 3), "dv": slice(2 * d // 3, d)}
        errs = {k: rel_err(dqkv[..., sl], dr[..., sl]) for k, sl in parts.items()}
        extra = dict(errs)
        if time_it:
            extra["us"] = round(timeit(lambda: native.attention_bwd(dout, qkv, out, lse, H)), 1)
        record(f"attention_bwd {tag}", max(errs.values()) < 3e-2, max(errs.values()), extra)
    run_case(f"attention {tag}", run)


def check_layernorm(M, D, time_it):
    tag = f"M{M} D{D}"
    x = (torch.randn(M, D, device=DEV) * 2 + 0.3).to(torch.bfloat16)
    dy = torch.randn(M, D, device=DEV).to(torch.bfloat16)
    gamma = torch.rand(D, device=DEV) + 0.5
    beta = torch.randn(D, device=DEV) * 0.1

    def run():
        y, mean, rstd = native.layernorm_fwd(x, gamma, beta, 1e-6)
        yr, mr, rr = reference.layernorm_fwd(x, gamma, beta, 1e-6)
        e = max(rel_err(y, yr), rel_err(mean, mr), rel_err(rstd, rr))
        extra = None
        if time_it:
            us = timeit(lambda: native.layernorm_fwd(x, gamma, beta, 1e-6))
            extra = {"us": round(us, 1), "GBps": round(4 * x.numel() / us / 1e3, 1)}
        record(f"layernorm_fwd {tag}", e < 1e-2, e, extra)
        dx, dg, db = native.layernorm_bwd(dy, x, mr, rr, gamma)
        dxr, dgr, dbr = reference.layernorm_bwd(dy, x, mr, rr, gamma)
        e = max(rel_err(dx, dxr), rel_err(dg, dgr), rel_err(db, dbr))
        extra = None
        if time_it:
            us = timeit(lambda: native.layernorm_bwd(dy, x, mr, rr, gamma))
            extra = {"us": round(us, 1), "GBps": round(6 * x.numel() / us / 1e3, 1)}
        record(f"layernorm_bwd {tag}", e < 1e-2, e, extra)
    run_case(f"layernorm {tag}", run)


def check_colsum(M, N, time_it):
    tag = f"M{M} N{N}"
    x = (torch.randn(M, N, device=DEV) + 0.25).to(torch.bfloat16)

    def run():
        got = native.colsum(x)
        want = x.double().sum(0).float()
        e = float((got - want).abs().max() / want.abs().max().clamp_min(1e-6))
        extra = None
        if time_it:
            us = timeit(lambda: native.colsum(x))
            us_t = timeit(lambda: x.sum(0, dtype=torch.float32))
            extra = {"us": round(us, 1), "GBps": round(2 * x.numel() / us / 1e3, 1), "aten_us": round(us_t, 1)}
        record(f"colsum {tag}", got.dtype == torch.float32 and e < 1e-4, e, extra)
    run_case(f"colsum {tag}", run)


# ----------------------------------------------------------------------------- others
def check_pool():
    for (B, C, H, W) in ((2, 64, 16, 16), (3, 64, 15, 17), (4, 64, 112, 112), (9, 64, 112, 112)):  # last: > 512 blocks (XCD grouping)
        def mp():
            x = nhwc(B, C, H, W).relu_().contiguous(memory_format=torch.channels_last)
            y, arg = native.maxpool3x3s2_fwd(x)
            yr, _ = reference.maxpool3x3s2_fwd(x)
            e = rel_err(y, yr)
            dy = nhwc(*y.shape)
            dx = native.maxpool3x3s2_bwd(dy, x, y, arg)
            dxr = reference.maxpool3x3s2_bwd(dy, x, yr)
            record(f"maxpool B{B} C{C} {H}x{W}", e == 0 and rel_err(dx, dxr) < 1e-2, max(e, rel_err(dx, dxr)))
        run_case("maxpool", mp)
    for (B, C, H, W) in ((4, 2048, 7, 7), (3, 512, 4, 4)):
        def ap():
            x = nhwc(B, C, H, W)
            y = native.global_avgpool_fwd(x)
            e = rel_err(y, reference.global_avgpool_fwd(x))
            dy = torch.randn(B, C, device=DEV).bfloat16()
            dx = native.global_avgpool_bwd(dy, x.shape)
            e2 = rel_err(dx, reference.global_avgpool_bwd(dy, x.shape))
            record(f"avgpool B{B} C{C} {H}x{W}", max(e, e2) < 1e-2, max(e, e2))
        run_case("avgpool", ap)


def check_stem_tail(B, C, H, W, time_it):
    """fused maxpool(relu(bn(u))) forward against the unfused native kernels (bn_apply_act -> maxpool): pooled
    values, winning taps and the ReLU mask must be IDENTICAL"""
    tag = f"B{B} C{C} {H}x{W}"

    def run():
        u = (nhwc(B, C, H, W) * 1.5 + 0.2).contiguous(memory_format=torch.channels_last)
        gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
        mean, invstd = reference.bn_finalize(reference.bn_stats(u), float(B * H * W), 1e-5)
        y, arg, mask = native.bn_relu_maxpool_fwd(u, mean, invstd, gamma, beta)
        z, act = native.bn_apply_act(u, mean, invstd, gamma, beta, None, True)
        y2, arg2 = native.maxpool3x3s2_fwd(z)
        e = float((y.float() != y2.float()).sum()) + float((arg != arg2).sum()) + float((mask != act).sum())
        extra = {}
        if time_it:
            extra["us_fused"] = round(timeit(lambda: native.bn_relu_maxpool_fwd(u, mean, invstd, gamma, beta)), 1)
            extra["us_unfused"] = round(timeit(lambda: native.maxpool3x3s2_fwd(
                native.bn_apply_act(u, mean, invstd, gamma, beta, None, True)[0])), 1)
        record(f"bn_relu_maxpool {tag}", e == 0, e, extra)
    run_case(f"bn_relu_maxpool {tag}", run)


def check_loss(time_it):
    for (B, N, dt) in ((256, 1000, torch.bfloat16), (33, 1000, torch.float32), (7, 10, torch.bfloat16), (5, 3001, torch.float32)):
        def ce():
            logits = (torch.randn(B, N, device=DEV) * 3).to(dt)
            labels = torch.randint(0, N, (B,), device=DEV)
            loss, dl = native.softmax_ce_fwd_bwd(logits, labels)
            lr, dlr = reference.softmax_ce_fwd_bwd(logits, labels)
            e = max(abs(float(loss) - float(lr)) / max(abs(float(lr)), 1e-6), rel_err(dl, dlr))
            extra = {"us": round(timeit(lambda: native.softmax_ce_fwd_bwd(logits, labels)), 1)} if time_it else None
            record(f"softmax_ce B{B} N{N} {dt}", e < 1e-2, e, extra)
        run_case("softmax_ce", ce)


def check_optim(time_it):
    def run(kind):
        g = torch.Generator(device="cpu").manual_seed(0)
        shapes = [(64, 3, 7, 7), (64,), (256, 64, 1, 1), (1000, 2048), (1000,), (513,), (128, 128, 3, 3)]
        ps = [torch.randn(s, generator=g).to(DEV) for s in shapes]
        gs = [torch.randn(s, generator=g).to(DEV) * 0.1 for s in shapes]
        lps = [p.to(torch.bfloat16) for p in ps]
        ps_r = [p.clone() for p in ps]
        lps_r = [p.clone() for p in lps]
        if kind == "adam":
            ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
            ms_r, vs_r = [m.clone() for m in ms], [v.clone() for v in vs]
            for step in (1, 2, 3):
                native.multi_adam(ps, gs, ms, vs, lps, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, True)
                for p, gg, m, v, lp in zip(ps_r, gs, ms_r, vs_r, lps_r):
                    reference.adam_step(p, gg, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.01, float(step), True, 1.0, lp)
        else:
            bufs = [torch.zeros_like(p) for p in ps]
            bufs_r = [b.clone() for b in bufs]
            for step in (1, 2, 3):
                native.multi_sgd(ps, gs, bufs, lps, 0.1, 0.9, 0.0, 1e-4, False, step == 1)
                for p, gg, b, lp in zip(ps_r, gs, bufs_r, lps_r):
                    reference.sgd_step(p, gg, b, 0.1, 0.9, 0.0, 1e-4, False, step == 1, 1.0, lp)
        e = max(rel_err(a, b) for a, b in zip(ps, ps_r))
        e_lp = max(rel_err(a, b) for a, b in zip(lps, lps_r))  # bf16 copies: may differ by one rounding step
        record(f"multi_{kind}", e < 1e-5 and e_lp < 1e-2, e, {"lp_err": e_lp})
    run_case("multi_adam", lambda: run("adam"))
    run_case("multi_sgd", lambda: run("sgd"))

    def flat():
        srcs = [torch.randn(n, device=DEV) for n in (5, 4096, 4097, 100000, 64)]
        for dt in (torch.float32, torch.bfloat16):
            dsts = [torch.zeros(n, device=DEV, dtype=dt) for n in (5, 4096, 4097, 100000, 64)]
            native.flatten_scaled_views(srcs, dsts, 0.125)
            e = max(rel_err(d, s * 0.125) for s, d in zip(srcs, dsts))
            record(f"multi_flatten -> {dt}", e < 1e-2, e)
    run_case("multi_flatten", flat)

    def cast():
        x = torch.rand(5, 3, 32, 32, device=DEV)
        y = native.nchw_to_nhwc_lowp(x)
        yr = reference.nchw_to_nhwc_lowp(x)
        ok = y.is_contiguous(memory_format=torch.channels_last)
        record("to_nhwc_bf16 f32 32x32", ok and rel_err(y, yr) == 0, rel_err(y, yr))
        x = torch.rand(3, 3, 224, 224, device=DEV)
        y = native.nchw_to_nhwc_lowp(x)
        record("to_nhwc_bf16 f32 224x224", rel_err(y, reference.nchw_to_nhwc_lowp(x)) == 0, rel_err(y, reference.nchw_to_nhwc_lowp(x)))
        for shp in ((2, 3, 15, 17), (2, 5, 9, 13), (130, 3, 64, 64)):  # LDS-tile kernel (odd sizes, C != 3); > 2^16 groups
            x = torch.rand(*shp, device=DEV) * 4 - 2
            y = native.nchw_to_nhwc_lowp(x, torch.bfloat16, 0.5)
            yr = reference.nchw_to_nhwc_lowp(x, torch.bfloat16, 0.5)
            record(f"to_nhwc_bf16 f32 {shp}", rel_err(y, yr) == 0, rel_err(y, yr))
        u = torch.randint(0, 256, (4, 32, 32, 3), device=DEV, dtype=torch.uint8)
        y = native.nchw_to_nhwc_lowp(u, torch.bfloat16, 1 / 255.0)
        record("to_nhwc_bf16 u8", rel_err(y, reference.nchw_to_nhwc_lowp(u, torch.bfloat16, 1 / 255.0)) < 1e-2,
               rel_err(y, reference.nchw_to_nhwc_lowp(u, torch.bfloat16, 1 / 255.0)))
    run_case("cast", cast)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="gpurun_out/kernel_check.json")
    a = ap.parse_args()
    only = set(a.only.split(",")) if a.only else None
    torch.manual_seed(0)
    t0 = time.time()
    if only is None or "elt" in only:
        check_optim(a.time)
        check_loss(a.time)
        check_pool()
    if only is None or "stemtail" in only or "bn" in only:
        for shp in ((2, 64, 16, 16), (3, 64, 15, 17), (2, 128, 9, 9)) + (() if a.quick else ((128, 64, 112, 112),)):
            check_stem_tail(*shp, a.time)
    if only is None or "bn" in only:
        shapes = [(4, 64, 8, 8), (3, 256, 7, 5), (2, 2048, 3, 3), (5, 128, 9, 9)]
        if not a.quick:
            shapes += [(128, 64, 56, 56), (128, 256, 56, 56), (128, 512, 28, 28), (128, 2048, 7, 7)]
        for s in shapes:
            check_bn(*s, a.time)
    if only is None or "attn" in only:
        for shp in ((2, 32, 1), (3, 197, 2), (2, 50, 3), (1, 256, 2), (2, 17, 1)) + (() if a.quick else ((128, 197, 12),)):
            check_attention(*shp, a.time)
    if only is None or "ln" in only:
        for shp in ((5, 768), (197 * 3 + 1, 256), (1030, 1024)) + (() if a.quick else ((128 * 197, 768),)):
            check_layernorm(*shp, a.time)
    if only is None or "ln" in only or "colsum" in only:
        for shp in ((1, 8), (37, 72), (1000, 768), (513, 1000)) + (() if a.quick else ((128 * 197, 768), (128 * 197, 2304), (128 * 197, 3072))):
            check_colsum(*shp, a.time)
    if only is None or "stem" in only:
        for shp in ((2, 32, 32), (3, 64, 48)) + (() if a.quick else ((128, 224, 224),)):
            check_stem(*shp, a.time)
    if only is None or "conv" in only:
        for c in conv_cases(a.quick):
            check_conv(c, a.time)
    n_fail = sum(1 for r in RESULTS if not r["ok"])
    print(f"\n{len(RESULTS)} cases, {n_fail} failed, {time.time() - t0:.1f}s", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"results": RESULTS, "failed": n_fail}, f, indent=1)
    return 1 if n_fail else 0  # 1 = numeric mismatches only (every case ran)


if __name__ == "__main__":
    sys.exit(main())
