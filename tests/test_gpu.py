"""T1/T3 on a real MI355X (``pytest -m gpu``): HIP kernels vs PyTorch fp32 oracles, the native
RCCL communicator and reducer at W=1, and the end-to-end script."""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

import pytorch_distributed_training_amd as pdt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kc():
    pdt.require_native()  # fail loudly: no silent eager fallback on a GPU box
    sys.path.insert(0, os.path.join(ROOT, "bench"))
    import kernel_check

    torch.manual_seed(0)
    return kernel_check


def _assert_all_ok(kc, start):
    bad = [r for r in kc.RESULTS[start:] if not r["ok"]]
    assert not bad, bad


def test_extension_is_loaded_from_tree():
    ext = pdt.require_native()
    assert os.path.dirname(ext.__file__) == os.path.join(ROOT, "pytorch_distributed_training_amd")
    assert ext.arch == "gfx950"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def test_elementwise_loss_pool_optimizer_kernels(kc):
    n = len(kc.RESULTS)
    kc.check_optim(False)
    kc.check_loss(False)
    kc.check_pool()
    _assert_all_ok(kc, n)


@pytest.mark.parametrize("shape", [(4, 64, 8, 8), (3, 256, 7, 5), (2, 2048, 3, 3), (32, 128, 28, 28)])
def test_batchnorm_kernels(kc, shape):
    n = len(kc.RESULTS)
    kc.check_bn(*shape, False)
    _assert_all_ok(kc, n)


@pytest.mark.parametrize("case", [
    (2, 64, 8, 8, 64, 1, 1, 0), (2, 64, 9, 7, 128, 3, 1, 1), (3, 128, 10, 10, 64, 3, 2, 1),
    (2, 256, 8, 8, 512, 1, 2, 0), (2, 512, 7, 7, 512, 3, 1, 1), (16, 64, 56, 56, 256, 1, 1, 0),
    (16, 128, 28, 28, 128, 3, 1, 1), (16, 256, 29, 29, 256, 3, 2, 1)])
def test_conv_kernels_fwd_dgrad_wgrad(kc, case):
    n = len(kc.RESULTS)
    kc.check_conv(case, False)
    _assert_all_ok(kc, n)


# Every implicit-GEMM build the headline benchmark runs at B = 1280, forced through the bindings' explicit ``cfg``
# argument on shapes small enough for the test budget but with ragged M tails (M % 256 != 0) and several N tiles:
# 10 = <256,128,32,2,3> (long-K / 256-channel 3x3), 15 = <256,64,32,2,3> (64-wide outputs), 6 = the 4-stage ring,
# 11 = <128,128|64,32,4,2> (short-K, with the fused BN-reduce epilogue in dgrad_bnr), 12 / 8 = the 3-stage rings.
@pytest.mark.parametrize("cfg,case", [
    (10, (3, 256, 15, 15, 256, 3, 1, 1)), (10, (5, 1024, 9, 9, 256, 1, 1, 0)), (10, (2, 128, 29, 29, 384, 3, 2, 1)),
    (15, (3, 64, 30, 30, 64, 3, 1, 1)), (15, (5, 256, 23, 23, 64, 1, 1, 0)), (15, (2, 128, 17, 17, 192, 3, 1, 1)),
    (6, (3, 256, 15, 15, 256, 3, 1, 1)), (6, (2, 512, 7, 7, 512, 3, 1, 1)), (6, (2, 320, 11, 11, 64, 3, 2, 1)),
    (11, (5, 64, 23, 23, 256, 1, 1, 0)), (11, (5, 256, 19, 19, 64, 1, 1, 0)), (11, (3, 128, 21, 21, 512, 1, 1, 0)),
    (11, (3, 128, 20, 20, 128, 3, 2, 1)), (12, (3, 64, 30, 30, 64, 3, 1, 1)), (8, (3, 128, 21, 21, 128, 3, 1, 1)),
    (8, (2, 512, 13, 13, 1024, 1, 2, 0))])
def test_conv_every_igemm_build_forced(kc, cfg, case):
    n = len(kc.RESULTS)
    kc.check_conv(case, False, cfg=cfg)
    _assert_all_ok(kc, n)


# Pointwise stride-1 convolutions on the GEMM core (gemm.hip: 256x256 tiles, persistent grid), which the bindings pick
# when the output channels fill its tile: forced here also on narrow / ragged shapes (M % 256 != 0, N = 64 .. 640,
# K = 64 .. 2048): forward + BN statistics, data gradient plain / + gated addend / + fused BN-backward sums (against
# the implicit-GEMM kernel and the separate reduce kernel).
@pytest.mark.parametrize("case", [
    (2, 64, 8, 8, 64, 1, 1, 0), (5, 64, 23, 23, 256, 1, 1, 0), (5, 256, 19, 19, 64, 1, 1, 0),
    (3, 128, 21, 21, 512, 1, 1, 0), (2, 512, 7, 7, 2048, 1, 1, 0), (2, 2048, 7, 7, 512, 1, 1, 0),
    (9, 512, 14, 14, 640, 1, 1, 0), (16, 256, 56, 56, 64, 1, 1, 0), (16, 64, 56, 56, 256, 1, 1, 0),
    # 3x3 / stride 1 / pad 1 in the core's spatial-filter mode (taps in the DMA's scalar offset, padding as per-row bits)
    (3, 64, 9, 7, 128, 3, 1, 1), (2, 256, 14, 14, 256, 3, 1, 1), (3, 512, 7, 7, 512, 3, 1, 1), (2, 64, 30, 30, 64, 3, 1, 1),
    (5, 128, 21, 21, 256, 3, 1, 1), (1, 256, 5, 3, 512, 3, 1, 1)])
def test_pointwise_conv_on_gemm_core_forced(kc, case):
    n = len(kc.RESULTS)
    kc.check_conv(case, False, cfg=kc.GEMM_CORE)
    _assert_all_ok(kc, n)


# 64 -> 64 channel 3x3 on the window kernel (weights in registers, input window staged once per tile): forced, and
# against the implicit-GEMM kernel for the fused BN-backward sums; ragged last row block (H % 4 != 0), more tiles
# than workgroups, the widest supported row (56), the narrowest.
@pytest.mark.parametrize("case", [
    (2, 64, 8, 8, 64, 3, 1, 1), (3, 64, 13, 13, 64, 3, 1, 1), (2, 64, 30, 30, 64, 3, 1, 1), (16, 64, 56, 56, 64, 3, 1, 1),
    (300, 64, 7, 5, 64, 3, 1, 1), (2, 64, 3, 2, 64, 3, 1, 1), (5, 64, 56, 17, 64, 3, 1, 1)])
def test_conv3x3_64ch_window_kernel_forced(kc, case):
    n = len(kc.RESULTS)
    kc.check_conv(case, False, cfg=kc.WIN64)
    _assert_all_ok(kc, n)


@pytest.mark.parametrize("shape", [(2, 8, 8), (3, 13, 16), (2, 30, 24), (16, 56, 56), (300, 7, 8), (2, 3, 8), (5, 56, 40)])
def test_wgrad_64ch_3x3_window_kernel_forced(shape):
    """wgrad_win64.hip (TN product out of an LDS window, all nine taps per staged tile, persistent grid) on shapes the
    work-size rule would leave to the per-tap kernel: partial last tiles (H % 4 != 0), more tiles than CUs, against
    the fp32 reference and the per-tap kernel."""
    from pytorch_distributed_training_amd import _C
    from pytorch_distributed_training_amd.ops import native, reference

    B, H, W = shape
    torch.manual_seed(2)
    mk = lambda: torch.randn(B, 64, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x, dy = mk(), mk()
    try:
        _C.set_wgrad_win64_mode(0)
        dw0 = native.conv2d_wgrad(dy, x, (64, 64, 3, 3), 1, 1)
        _C.set_wgrad_win64_mode(2)
        dw1 = native.conv2d_wgrad(dy, x, (64, 64, 3, 3), 1, 1)
    finally:
        _C.set_wgrad_win64_mode(-1)
    ref = reference.conv2d_wgrad(dy.float(), x.float(), (64, 64, 3, 3), 1, 1)
    scale = ref.abs().max()
    assert (dw1.float() - ref).abs().max() <= 5e-3 * scale
    assert (dw1.float() - dw0.float()).abs().max() <= 5e-3 * scale


@pytest.mark.parametrize("case,splits", [
    ((3, 128, 21, 21, 128, 3, 1, 1), 7), ((5, 256, 9, 9, 1024, 1, 1, 0), 5), ((2, 64, 30, 30, 64, 3, 1, 1), 13),
    ((3, 64, 23, 23, 128, 1, 1, 0), 9), ((3, 128, 20, 20, 64, 3, 2, 1), 4), ((2, 512, 7, 7, 512, 3, 1, 1), 3),
    ((3, 256, 11, 11, 128, 1, 1, 0), 3), ((2, 256, 13, 13, 128, 3, 1, 1), 5), ((3, 128, 18, 18, 256, 3, 2, 1), 4),
    # Cout % 256 == 0 and Cin % 256 == 0: the 256x256 LDS-DMA kernel (gemm_tn.hip): default and forced split counts
    # (ragged and empty last splits), 3x3 with padding, stride 2 (3x3 and 1x1), several tiles per operand
    ((6, 256, 9, 9, 256, 3, 1, 1), 0), ((6, 256, 9, 9, 256, 3, 1, 1), 7), ((4, 256, 14, 14, 512, 3, 2, 1), 3),
    ((4, 512, 12, 12, 256, 1, 2, 0), 2), ((3, 256, 11, 11, 768, 1, 1, 0), 11), ((5, 512, 7, 7, 512, 3, 1, 1), 0),
    ((2, 256, 8, 8, 256, 1, 1, 0), 4), ((16, 1024, 14, 14, 256, 1, 1, 0), 0)])
def test_conv_wgrad_every_tile_with_split_k(kc, case, splits):
    """k_conv_wgrad<128,128>, <128,64>, <64,128>, <64,64> with several splits (ragged last split) + the slab reduce"""
    n = len(kc.RESULTS)
    ext = pdt.require_native()
    ext.set_wgrad_tn_min_rows(128)  # (production: 8192 output pixels; these shapes are small on purpose)
    try:
        kc.check_conv(case, False, splits=splits)
    finally:
        ext.set_wgrad_tn_min_rows(8192)
    _assert_all_ok(kc, n)


@pytest.mark.parametrize("shape", [(2, 32, 1), (3, 197, 2), (2, 50, 3), (1, 256, 2), (2, 17, 1), (8, 197, 12)])
def test_attention_kernels(kc, shape):
    n = len(kc.RESULTS)
    kc.check_attention(*shape, False)
    _assert_all_ok(kc, n)


@pytest.mark.parametrize("shape", [(3, 197, 2), (40, 197, 12), (2, 129, 3), (25, 160, 12), (30, 192, 12), (23, 224, 12),
                                   (1, 256, 2), (24, 250, 12), (1, 161, 1)])
def test_attention_persistent_pipelined_kernels(kc, shape):
    """Whole-head attention of 5 to 8 tiles on the persistent kernels (LDS-DMA staging, next head in flight): fewer heads
    than CUs, several heads per workgroup (up to 480 heads on 256 CUs: uneven trip counts), full and partial last tiles;
    against fp32, and against the one-head-per-workgroup kernels (to bf16 rounding: exp2 of one FMA instead of
    expf of a difference, D[q] summed in a different order)."""
    from pytorch_distributed_training_amd.ops import native

    ext = pdt.require_native()
    n = len(kc.RESULTS)
    kc.check_attention(*shape, False)
    _assert_all_ok(kc, n)
    b, s_, h = shape
    g = torch.Generator(device="cuda").manual_seed(b * 1000 + s_)
    qkv = torch.randn(b, s_, 3 * h * 64, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(b, s_, h * 64, device="cuda", generator=g).to(torch.bfloat16)
    o1, l1 = native.attention_fwd(qkv, h)
    d1 = native.attention_bwd(do, qkv, o1, l1, h)
    try:
        ext.set_attention_pipelined(False)
        o0, l0 = native.attention_fwd(qkv, h)
        d0 = native.attention_bwd(do, qkv, o0, l0, h)
    finally:
        ext.set_attention_pipelined(True)
    assert ((o0.float() - o1.float()).abs().max() / o0.float().abs().max()) < 1e-2 and (l0 - l1).abs().max() < 1e-4
    assert ((d0.float() - d1.float()).abs().max() / d0.float().abs().max()) < 1e-2
    # a second pass over the same inputs gives the same bits (no dependence on which workgroup took which head)
    o2, l2 = native.attention_fwd(qkv, h)
    d2 = native.attention_bwd(do, qkv, o2, l2, h)
    assert torch.equal(o1, o2) and torch.equal(d1, d2)


@pytest.mark.parametrize("S,hd", [(197, 64), (130, 64), (64, 64), (300, 64), (197, 128)])
def test_attention_with_log_sum_exp_far_below_zero(S, hd):
    """Scores that are ALL strongly negative (lse < -88): exp(0 - lse) of a padded key's zero score overflows fp32.  The
    padded keys must not leak inf * 0 = NaN into dQ (the pipelined backward did, in a diverging ViT run:
    profiles/r3_convergence_vit.md); every kernel family against fp32."""
    from pytorch_distributed_training_amd.ops import native, reference

    ext = pdt.require_native()
    H = 2
    g = torch.Generator(device="cuda").manual_seed(S + hd)
    a = 3.5 * (64.0 / hd) ** 0.25  # q . k / sqrt(hd) ~ -98 for every pair
    q = a + 0.05 * torch.randn(3, S, H * hd, device="cuda", generator=g)
    k = -a + 0.05 * torch.randn(3, S, H * hd, device="cuda", generator=g)
    v = torch.randn(3, S, H * hd, device="cuda", generator=g)
    qkv = torch.cat([q, k, v], dim=-1).to(torch.bfloat16)
    do = torch.randn(3, S, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    outr, lser = reference.attention_fwd(qkv.float(), H)
    assert float(lser.max()) < -88.0
    dr = reference.attention_bwd(do.float(), qkv.float(), outr, lser, H)
    for pipelined in (True, False):
        ext.set_attention_pipelined(pipelined)
        try:
            out, lse = native.attention_fwd(qkv, H)
            d = native.attention_bwd(do, qkv, out, lse, H)
        finally:
            ext.set_attention_pipelined(True)
        assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(d.float()).all()), pipelined
        assert ((out.float() - outr).abs().max() / outr.abs().max()) < 2e-2
        assert (lse - lser).abs().max() < 1e-2
        assert ((d.float() - dr).abs().max() / dr.abs().max()) < 3e-2, pipelined


@pytest.mark.parametrize("shape", [(2, 257, 2, 64), (1, 577, 3, 64), (2, 300, 2, 32), (3, 65, 1, 32), (1, 1000, 2, 128),
                                   (2, 197, 2, 128), (70, 130, 1000, 32)])
def test_attention_block_streaming_kernels(kc, shape):
    """Any sequence length, head_dim 32 / 64 / 128: the key-/query-block kernels (beyond 256 tokens, or a head size the
    whole-head kernels are not built for); the last case cuts the batch (more than 65535 (batch, head) pairs)."""
    n = len(kc.RESULTS)
    b, s_, h, hd = shape
    kc.check_attention(b, s_, h, False, hd=hd)
    _assert_all_ok(kc, n)


def test_attention_block_streaming_equals_whole_head_kernels():
    """The two kernel families on the flagship shape (197 tokens, head_dim 64): same outputs and gradients to bf16
    rounding (PDT_ATTN_GENERIC=1 is read once per process, so the forced run is a child process)."""
    import subprocess
    import sys

    code = (
        "import torch, sys; sys.path.insert(0, %r)\n"
        "from pytorch_distributed_training_amd.ops import native\n"
        "g = torch.Generator().manual_seed(5)\n"
        "qkv = torch.randn(4, 197, 3 * 4 * 64, generator=g).cuda().bfloat16()\n"
        "do = torch.randn(4, 197, 4 * 64, generator=g).cuda().bfloat16()\n"
        "o, l = native.attention_fwd(qkv, 4); d = native.attention_bwd(do, qkv, o, l, 4)\n"
        "torch.save({'o': o.cpu(), 'l': l.cpu(), 'd': d.cpu()}, sys.argv[1])\n" % ROOT)
    outs = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, env in (("whole", {}), ("stream", {"PDT_ATTN_GENERIC": "1"})):
            f = os.path.join(td, tag + ".pt")
            subprocess.run([sys.executable, "-c", code, f], check=True, env={**os.environ, **env}, timeout=300)
            outs[tag] = torch.load(f, weights_only=True)
    for k, tol in (("o", 2e-2), ("l", 1e-3), ("d", 3e-2)):
        a, b = outs["whole"][k].float(), outs["stream"][k].float()
        assert ((a - b).abs().max() / a.abs().max().clamp_min(1e-6)) < tol, k


def _close(a, b, tol, what):
    a, b = a.float(), b.float()
    assert bool(torch.isfinite(a).all()), f"{what}: non-finite values"
    err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
    assert err < tol, (what, err)


def test_kernels_on_extreme_but_legal_inputs():
    """What random-normal test data never shows: constant rows (zero variance), a residual stream in the hundreds,
    logits far apart, a softmax with one overwhelming key, GELU far into both tails, channels that are constant over
    the batch.  Every output finite and equal to the fp32 reference of the same inputs."""
    from pytorch_distributed_training_amd.ops import native, reference

    g = torch.Generator(device="cuda").manual_seed(77)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    # LayerNorm: constant rows, a large offset with small spread, values near the top of what bf16 activations reach
    x = rn(300, 768)
    x[0:8] = 3.0
    x[8:16] = 250.0 + 0.5 * rn(8, 768)
    x[16:24] = 3.0e4 * rn(8, 768)
    x = x.to(torch.bfloat16)
    gamma, beta = torch.rand(768, device="cuda", generator=g) + 0.5, 0.1 * rn(768)
    y, mean, rstd = native.layernorm_fwd(x, gamma, beta, 1e-6)
    yr, mr, rr = reference.layernorm_fwd(x, gamma, beta, 1e-6)
    _close(y, yr, 2e-2, "layernorm y")
    _close(rstd, rr, 1e-3, "layernorm rstd")
    dy, add = rn(300, 768).to(torch.bfloat16), rn(300, 768).to(torch.bfloat16)
    for got, want, what in zip(native.layernorm_bwd(dy, x, mr, rr, gamma, add), reference.layernorm_bwd(dy, x, mr, rr, gamma, add),
                               ("dx", "dgamma", "dbeta")):
        _close(got, want, 2e-2, "layernorm " + what)
    # softmax cross-entropy: logits hundreds apart, the label on the smallest one
    logits = (40.0 * rn(64, 1000)).to(torch.bfloat16)
    labels = logits.float().argmin(dim=1)
    loss, dl = native.softmax_ce_fwd_bwd(logits, labels)
    lossr, dlr = reference.softmax_ce_fwd_bwd(logits, labels)
    _close(loss, lossr, 1e-3, "softmax-CE loss")
    _close(dl, dlr, 2e-2, "softmax-CE gradient")
    # attention: one key that dwarfs the rest (scores ~ +-400), and keys / queries of very different norms
    qkv = rn(2, 197, 3 * 2 * 64)
    qkv[:, :, :128] *= 6.0
    qkv[:, 5, 128:256] *= 8.0
    qkv = qkv.to(torch.bfloat16)
    do = rn(2, 197, 128).to(torch.bfloat16)
    out, lse = native.attention_fwd(qkv, 2)
    outr, lser = reference.attention_fwd(qkv.float(), 2)
    _close(out, outr, 2e-2, "attention out")
    _close(lse, lser, 1e-3, "attention lse")
    _close(native.attention_bwd(do, qkv, out, lse, 2), reference.attention_bwd(do.float(), qkv.float(), outr, lser, 2), 4e-2,
           "attention dqkv")
    # GEMM epilogues: GELU and GELU' far into both tails, a bias that dominates the product
    a = (3.0 * rn(512, 256)).to(torch.bfloat16)
    w = rn(512, 256).to(torch.bfloat16)
    bias = 50.0 * rn(512)
    c, pre, _ = native.gemm_nt(a, w, bias=bias, act=1, want_preact=True)
    cr, prer, _ = reference.gemm_nt(a, w, bias=bias, act=1, want_preact=True)
    _close(pre, prer, 2e-2, "gemm pre-activation")
    _close(c, cr, 2e-2, "gemm GELU")
    u = (60.0 * rn(512, 512)).to(torch.bfloat16)
    d, _, cs = native.gemm_nt(a, w, gelu_in=u, want_colsum=True)
    dr_, _, csr = reference.gemm_nt(a, w, gelu_in=u, want_colsum=True)
    _close(d, dr_, 2e-2, "gemm x GELU'")
    _close(cs, csr, 2e-2, "gemm column sums")
    # BatchNorm: a channel that is constant over the batch, a channel with a huge offset
    xb = rn(8, 64, 14, 14)
    xb[:, 3] = 1.25
    xb[:, 5] += 1000.0
    xb = xb.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    st = native.bn_stats(xb)
    mean_, invstd = native.bn_finalize(st, float(8 * 14 * 14), 1e-5)
    meanr, invstdr = reference.bn_finalize(reference.bn_stats(xb), float(8 * 14 * 14), 1e-5)
    _close(mean_, meanr, 1e-3, "bn mean")
    assert bool(torch.isfinite(invstd).all()) and float(invstd[3]) == pytest.approx(1e-5 ** -0.5, rel=1e-2)
    # (channel 5: var 1 next to mean 1000 in bf16 steps of 8 -- both paths see the same rounded data)
    _close(invstd, invstdr, 2e-2, "bn invstd")
    one, zero = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    yb = native.bn_apply_act(xb, mean_, invstd, one, zero, relu=True)
    ybr = reference.bn_apply_act(xb, meanr, invstdr, one, zero, relu=True)
    _close(yb[0] if isinstance(yb, tuple) else yb, ybr[0] if isinstance(ybr, tuple) else ybr, 3e-2, "bn apply")


@pytest.mark.parametrize("shape", [(5, 768), (592, 256), (1030, 1024), (8 * 197, 768), (77, 32), (300, 192), (129, 384),
                                   (64, 1536), (33, 2048), (10, 100)])
def test_layernorm_kernels(kc, shape):
    n = len(kc.RESULTS)
    kc.check_layernorm(*shape, False)
    _assert_all_ok(kc, n)


@pytest.mark.parametrize("shape", [(1, 8), (37, 72), (1000, 768), (513, 1000), (8 * 197, 3072)])
def test_colsum_kernel(kc, shape):
    """bias-gradient column sums (bf16 rows -> fp32) against a float64 sum"""
    n = len(kc.RESULTS)
    kc.check_colsum(*shape, False)
    _assert_all_ok(kc, n)


@pytest.fixture(scope="module")
def gc():
    pdt.require_native()
    sys.path.insert(0, os.path.join(ROOT, "bench", "micro"))
    import gemm_check

    return gemm_check


@pytest.mark.parametrize("mnk", [(256, 256, 64), (300, 264, 128), (17, 8, 72), (513, 1000, 2048), (1000, 512, 1000),
                                 (2 * 197, 2304, 768), (70000, 256, 192)])
@pytest.mark.parametrize("bias", [False, True])
def test_gemm_plain_and_bias(gc, mnk, bias):
    """csrc/kernels/gemm.hip vs fp32 matmul: ragged M / N / K, several tiles per persistent workgroup (70000 rows =
    274 tiles on 256 CUs), per-element relative error"""
    assert gc.check(*mnk, bias=bias, timing=False)


@pytest.mark.parametrize("kw", [dict(bias=True, act=1, preact=True), dict(bias=True, residual=True),
                                dict(gelu_in=True, colsum=True), dict(colsum=True), dict(residual=True, colsum=True)])
@pytest.mark.parametrize("mnk", [(777, 520, 192), (66000, 768, 128)])
def test_gemm_fused_epilogues(gc, mnk, kw):
    """GELU + pre-activation, residual add, GELU' multiply and the column-sum reduction of the GEMM epilogue"""
    assert gc.check(*mnk, timing=False, **kw)


@pytest.fixture()
def narrow_tiles():
    """every GEMM-core launch inside the test on the 256 x 128 tile build (gemm.hip, BNT = 128)"""
    ext = pdt.require_native()
    ext.set_gemm_tile_width(128)
    yield ext
    ext.set_gemm_tile_width(0)


@pytest.mark.parametrize("kw", [dict(), dict(bias=True), dict(bias=True, act=1, preact=True), dict(bias=True, residual=True),
                                dict(gelu_in=True, colsum=True), dict(colsum=True), dict(residual=True, colsum=True)])
@pytest.mark.parametrize("mnk", [(777, 520, 192), (66000, 768, 128), (300, 128, 1000), (513, 1000, 2048)])
def test_gemm_128_wide_tile_build(gc, narrow_tiles, mnk, kw):
    """The 128-wide tile build of the GEMM core (no B-hi half-tile, 8 x 2 accumulators per wave, 32-row store passes)
    on every epilogue: ragged M / N / K (K % 64 != 0 only without a fused operand, as in the square build), more tiles
    than workgroups (66000 rows x 6 tile columns = 1548 tiles)."""
    if mnk[2] % 64 != 0 and (kw.get("act") or kw.get("residual") or kw.get("gelu_in")):
        pytest.skip("ragged K takes no fused operand (launch_gemm_nt)")
    assert gc.check(*mnk, timing=False, **kw)


@pytest.mark.parametrize("case", [
    (5, 64, 23, 23, 128, 1, 1, 0), (5, 256, 19, 19, 128, 1, 1, 0), (3, 128, 21, 21, 384, 1, 1, 0), (16, 512, 28, 28, 128, 1, 1, 0),
    (3, 128, 9, 7, 128, 3, 1, 1), (5, 128, 28, 28, 128, 3, 1, 1), (2, 256, 14, 14, 256, 3, 1, 1), (1, 64, 5, 3, 640, 3, 1, 1)])
def test_conv_on_gemm_core_128_wide_tiles(kc, narrow_tiles, case):
    """Pointwise and 3x3 convolutions forced onto the GEMM core with its 128-wide tiles: forward + BN statistics, data
    gradient plain / + gated addend / + fused BN-backward sums, against fp32 and the implicit-GEMM kernel."""
    n = len(kc.RESULTS)
    kc.check_conv(case, False, cfg=kc.GEMM_CORE)
    _assert_all_ok(kc, n)


def test_gemm_tile_width_choice_gives_identical_products():
    """256- and 128-wide tiles sum every output element over the same K order: bit-identical results"""
    ext = pdt.require_native()
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(3000, 320, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.randn(768, 320, device="cuda", generator=g).to(torch.bfloat16)
    try:
        ext.set_gemm_tile_width(256)
        c256 = ext.gemm_nt(a, b)[0]
        ext.set_gemm_tile_width(128)
        c128 = ext.gemm_nt(a, b)[0]
    finally:
        ext.set_gemm_tile_width(0)
    assert torch.equal(c256, c128)


def test_gemm_identity_weight_asymmetric_input():
    """A = asymmetric integers, B = I: the output must be A (catches a transposed accumulator map or swapped operands)"""
    ext = pdt.require_native()
    m, k = 300, 256
    a = (torch.arange(m * k, device="cuda", dtype=torch.float32).reshape(m, k) % 251 - 125).to(torch.bfloat16)
    eye = torch.eye(k, device="cuda", dtype=torch.bfloat16)
    c = ext.gemm_nt(a, eye)[0]
    assert torch.equal(c, a)


def test_linear_wgrad_in_row_chunks_keeps_the_wide_kernel_full():
    """A token axis longer than one buffer descriptor (ViT at a batch sized for 288 GB) goes through linear_wgrad in row
    chunks: each chunk is its own launch of the 256 x 256 TN kernel with its full split count (round 3: the splits were
    divided among the chunks and every launch filled half the CUs).  Same result as the single launch and as fp32."""
    ext = pdt.require_native()
    g = torch.Generator(device="cuda").manual_seed(21)
    dy = torch.randn(20000, 512, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(20000, 256, device="cuda", generator=g).to(torch.bfloat16)
    ref = dy.float().t() @ x.float()
    ext.set_wgrad_tn_min_rows(1024)
    try:
        one = ext.linear_wgrad(dy, x)
        ext.set_activation_window(8 << 20)  # 8192 rows of 1 KiB per launch: three chunks
        try:
            cut = ext.linear_wgrad(dy, x)
        finally:
            ext.set_activation_window((1 << 31) - 4096)
    finally:
        ext.set_wgrad_tn_min_rows(8192)
    scale = ref.abs().mean()
    assert ((one - ref).abs() / (ref.abs() + scale)).max() < 1e-3
    assert ((cut - ref).abs() / (ref.abs() + scale)).max() < 1e-3
    assert ((cut - one).abs() / (one.abs() + scale)).max() < 1e-5


def test_linear_wgrad_and_vit_glue_kernels():
    ext = pdt.require_native()
    from pytorch_distributed_training_amd.ops import native, reference

    g = torch.Generator(device="cuda").manual_seed(3)
    dy = torch.randn(3 * 197, 1000, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(3 * 197, 768, device="cuda", generator=g).to(torch.bfloat16)
    dw = native.linear_wgrad(dy, x)
    ref = dy.float().t() @ x.float()
    assert dw.dtype == torch.float32 and ((dw - ref).abs() / (ref.abs() + ref.abs().mean())).max() < 1e-3
    # both widths multiples of 256 and >= 8192 tokens: the 256x256 LDS-DMA kernel (gemm_tn.hip), ragged last split
    dy = torch.randn(9001, 768, device="cuda", generator=g).to(torch.bfloat16)[:9000]
    x = torch.randn(9000, 1024, device="cuda", generator=g).to(torch.bfloat16)
    dw = native.linear_wgrad(dy, x)
    ref = dy.float().t() @ x.float()
    assert ((dw - ref).abs() / (ref.abs() + ref.abs().mean())).max() < 1e-3
    img = torch.randn(3, 3, 64, 64, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert torch.equal(native.patchify(img, 16), reference.patchify(img, 16))
    tok = torch.randn(5, 16, 768, device="cuda", generator=g).to(torch.bfloat16)
    cls, pos = torch.randn(1, 1, 768, device="cuda", generator=g), torch.randn(1, 17, 768, device="cuda", generator=g)
    assert (native.vit_embed_fwd(tok, cls, pos).float() - reference.vit_embed_fwd(tok, cls, pos).float()).abs().max() < 2e-2
    dxx = torch.randn(19, 17, 768, device="cuda", generator=g).to(torch.bfloat16)
    dp, dpos = native.vit_embed_bwd(dxx)
    rp, rpos = reference.vit_embed_bwd(dxx)
    assert torch.equal(dp, rp) and (dpos - rpos).abs().max() < 1e-3
    # LayerNorm backward with the fused residual-gradient addend
    xx = torch.randn(100, 768, device="cuda", generator=g).to(torch.bfloat16)
    gam, bet = torch.randn(768, device="cuda", generator=g), torch.randn(768, device="cuda", generator=g)
    y, mean, rstd = native.layernorm_fwd(xx, gam, bet, 1e-6)
    dyy, add = torch.randn_like(xx), torch.randn_like(xx)
    dx1 = native.layernorm_bwd(dyy, xx, mean, rstd, gam, add)[0]
    dx2 = reference.layernorm_bwd(dyy, xx, mean, rstd, gam, add)[0]
    assert (dx1.float() - dx2.float()).abs().max() < 5e-2


def test_vit_native_matches_fp32_oracle_gradients():
    """One ViT step (2 layers, width 768) on the fused bf16 nodes against the same weights run through the fp32
    PyTorch oracles: loss and every parameter gradient must agree (cosine), none skipped."""
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models.vit import VisionTransformer
    from pytorch_distributed_training_amd.nn import CrossEntropyLoss

    torch.manual_seed(0)
    net = VisionTransformer(64, 16, 2, 12, 768, 3072, 1000).cuda()
    for p in net.parameters():  # zero-initialised heads / tokens would make half the gradients trivially zero
        if p.abs().sum() == 0:
            torch.nn.init.normal_(p, std=0.02)
    ref = VisionTransformer(64, 16, 2, 12, 768, 3072, 1000).cuda()
    ref.load_state_dict(net.state_dict())
    for p in net.parameters():
        p._lp = p.detach().to(torch.bfloat16)
        ref_p = dict(ref.named_parameters())
    for (n, p) in net.named_parameters():
        ref_p[n].data.copy_(p._lp.float())  # the oracle runs on the SAME bf16-rounded weights
    x = torch.randn(6, 3, 64, 64, device="cuda")
    y = torch.randint(0, 1000, (6,), device="cuda")
    crit = CrossEntropyLoss()
    xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    loss = crit(net(xb), y)
    loss.backward()
    ops.set_force_reference(True)
    try:
        loss_ref = crit(ref(xb.float()), y)
        loss_ref.backward()
    finally:
        ops.set_force_reference(False)
    assert abs(float(loss) - float(loss_ref)) < 3e-2 * max(1.0, abs(float(loss_ref)))
    for (n, p) in net.named_parameters():
        g, gr = p.grad.float().flatten(), ref_p[n].grad.float().flatten()
        cos = torch.nn.functional.cosine_similarity(g, gr, dim=0).item()
        assert cos > 0.98, (n, cos)
        assert 0.9 < (g.norm() / gr.norm().clamp_min(1e-12)).item() < 1.1, n


@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 96, 64), (2, 40, 32), (5, 224, 224), (300, 32, 32)])
def test_stem_conv_window_kernel_forced(kc, shape):
    """conv_stem_win.hip (weights in registers, 8 output rows per tile, persistent grid) on shapes the work-size rule
    would leave to the implicit-GEMM kernel, with partial last tiles (Ho % 8 != 0) and more tiles than CUs: against
    the fp32 reference (kernel_check) and against the implicit-GEMM launch of the same packed operands."""
    from pytorch_distributed_training_amd import _C
    from pytorch_distributed_training_amd.ops import native

    B, H, W = shape
    torch.manual_seed(1)
    x = torch.randn(B, 3, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, device="cuda") / 12).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    try:
        _C.set_stem_win_mode(0)
        y0, st0, _ = native.conv2d_fwd(x, w, 2, 3, True, True)
        _C.set_stem_win_mode(2)
        y1, st1, _ = native.conv2d_fwd(x, w, 2, 3, True, True)
        assert st1.shape[0] == min(B * ((H // 2 + 7) // 8), torch.cuda.get_device_properties(0).multi_processor_count)
        assert (y1.float() - y0.float()).abs().max() <= 1e-2 * y0.float().abs().max()
        s0, s1 = native.bn_sum_partials(st0), native.bn_sum_partials(st1)
        assert torch.allclose(s1, s0, rtol=2e-3, atol=2e-2 * float(s0.abs().max()) * 1e-1)
        dy = torch.randn(B, 64, H // 2, W // 2, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        xp = native.conv2d_fwd(x, w, 2, 3, False, True)[2]
        dw1 = native.conv2d_wgrad(dy, xp, (64, 3, 7, 7), 2, 3)
        _C.set_stem_win_mode(0)
        dw0 = native.conv2d_wgrad(dy, xp, (64, 3, 7, 7), 2, 3)
        _C.set_stem_win_mode(2)
        assert (dw1 - dw0).abs().max() <= 2e-3 * dw0.abs().max()
        n = len(kc.RESULTS)
        kc.check_stem(B, H, W, False)
        _assert_all_ok(kc, n)
    finally:
        _C.set_stem_win_mode(-1)


@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 64, 48), (4, 224, 224)])
def test_stem_conv_native_fwd_wgrad(kc, shape):
    n = len(kc.RESULTS)
    kc.check_stem(*shape, False)
    _assert_all_ok(kc, n)


def test_conv_identity_weight_asymmetric_input():
    """Layout check the guide asks for: W = I with an asymmetric input must come back unchanged
    (a transposed C/D mapping or swapped operand order cannot pass)."""
    from pytorch_distributed_training_amd.ops import native

    B, C, H, W = 2, 128, 5, 7
    x = torch.arange(B * C * H * W, device="cuda", dtype=torch.float32).reshape(B, C, H, W) % 251 - 125
    x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.eye(C, device="cuda").reshape(C, C, 1, 1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, _ = native.conv2d_fwd(x, w, 1, 0, False)
    assert torch.equal(y, x)
    dx = native.conv2d_dgrad(x, w, (B, C, H, W), 1, 0)
    assert torch.equal(dx, x)


def test_rccl_comm_world1_and_stream_fencing():
    ext = pdt.require_native()
    comm = ext.RcclComm(0, 1, ext.RcclComm.make_unique_id(), 0, True)
    t = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    want = t.clone() * 2
    t.mul_(2)               # compute stream work the collective must wait for
    comm.all_reduce_(t)     # world 1: identity, but goes through ncclAllReduce on the comm stream
    t2 = t + 0              # compute stream work that must wait for the collective
    torch.cuda.synchronize()
    assert torch.equal(t2, want)
    b = torch.full((1000,), 3.0, device="cuda", dtype=torch.bfloat16)
    comm.broadcast_(b, 0)
    comm.reduce_(b, 0)
    out = torch.empty(1000, device="cuda", dtype=torch.bfloat16)
    comm.all_gather_(b, out)
    comm.synchronize()
    assert torch.equal(out, b) and comm.async_error() == ""


def _engine(model="resnet50", dtype="bf16", **kw):
    from pytorch_distributed_training_amd import dist as pdist
    from pytorch_distributed_training_amd.engine import Engine, EngineConfig

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    if not pdist.is_initialized():
        pdist.init_process_group(backend="rccl", local_rank=0)
    return Engine(EngineConfig(model=model, num_classes=kw.pop("num_classes", 1000), dtype=dtype, seed=kw.pop("seed", 0), **kw))


def test_resnet50_bf16_step_loss_decreases_on_fixed_batch():
    from pytorch_distributed_training_amd.nn import functional as pF

    eng = _engine(optimizer="sgd", lr=0.01)
    assert eng.net.reducer_kind == "native"
    g = torch.Generator().manual_seed(0)
    x = torch.rand(16, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (16,), generator=g).cuda()
    losses = [float(eng.train_step(x, y)) for _ in range(8)]
    assert all(l == l for l in losses) and losses[-1] < losses[0] - 0.5, losses
    # gradients alias the reducer's flat buckets, master weights live in the arena, bf16 copies refreshed
    buckets = {t.untyped_storage().data_ptr() for t in eng.net.reducer.bucket_tensors()}
    assert all(p.grad.untyped_storage().data_ptr() in buckets for p in eng.net.parameters())
    w = eng.net.module.layer1[0].conv1.weight
    assert w.dtype == torch.float32 and w._lp.dtype == torch.bfloat16
    assert torch.equal(w._lp, w.detach().to(torch.bfloat16))
    assert not any(k.startswith("conv2d") for k in pF.fallback_counts), dict(pF.fallback_counts)  # stem included


@pytest.mark.parametrize("model,img,damp", [("resnet18", 64, 0.25), ("resnet50", 96, 0.1)])
def test_bf16_native_matches_oracles(model, img, damp):
    """The native bf16 HIP path against an fp32 run of the PyTorch oracles on the SAME bf16-rounded weights and
    inputs (so the only difference is where activations are rounded and how sums are ordered), and against the
    stock library ops on bf16 tensors.  EVERY parameter gradient is checked, none is skipped.

    The residual branches are damped (last BN weight of every block = ``damp``, the non-zero cousin of
    zero_init_residual): an undamped randomly initialised 50-layer BN network amplifies bf16 rounding chaotically
    (library bf16 and native both end 35 %% away from fp32 in the logits), which measures the network, not the
    kernels.  Measured with these settings: worst gradient cosine 0.94 (resnet50) / 0.93 (resnet18)."""
    from pytorch_distributed_training_amd import nn as pnn
    from pytorch_distributed_training_amd import ops
    from pytorch_distributed_training_amd.models import build_model
    from pytorch_distributed_training_amd.parallel import ParamArena

    torch.manual_seed(0)
    net = build_model(model, num_classes=10).cuda()
    last_bn = "bn2.weight" if model in ("resnet18", "resnet34") else "bn3.weight"
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith(last_bn):
                p.fill_(damp)
        for p in net.parameters():
            p.copy_(p.to(torch.bfloat16).float())  # the fp32 oracle runs on exactly the weights the bf16 path sees
    x = torch.randn(32, 3, img, img, device="cuda").to(torch.bfloat16)
    y = torch.randint(0, 10, (32,), device="cuda")
    crit = pnn.CrossEntropyLoss()

    def run(inp, reference):
        for p in net.parameters():
            p.grad = None
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
        ops.set_force_reference(reference)
        try:
            out = net(inp)
            loss = crit(out, y)
            loss.backward()
        finally:
            ops.set_force_reference(False)
        return out.detach().float(), float(loss.detach()), {n: p.grad.detach().float().clone() for n, p in net.named_parameters()}

    out_a, loss_a, g_a = run(x.float(), True)
    arena = ParamArena(net)
    arena.attach_lowp(torch.bfloat16)
    xb = x.contiguous(memory_format=torch.channels_last)
    out_b, loss_b, g_b = run(xb, True)
    out_c, loss_c, g_c = run(xb, False)

    def rel(u, v):
        return float((u - v).abs().max() / v.abs().max())

    def cos(u, v):
        u, v = u.flatten(), v.flatten()
        return float(torch.dot(u, v) / (u.norm() * v.norm() + 1e-12))

    eb, ec = rel(out_b, out_a), rel(out_c, out_a)
    print(f"{model}: logits err vs fp32: library-bf16 {eb:.4f}, native {ec:.4f}; native vs library {rel(out_c, out_b):.4f}")
    assert ec < 5e-2 and ec < 1.5 * eb + 1e-2
    assert abs(loss_c - loss_a) < 2e-2
    worst_b = worst_c = 1.0
    for n in g_a:  # no parameter is skipped
        cb, cc = cos(g_b[n], g_a[n]), cos(g_c[n], g_a[n])
        worst_b, worst_c = min(worst_b, cb), min(worst_c, cc)
        assert cc > 0.9, (n, cb, cc)
        assert cc > cb - 0.05, (n, cb, cc)
        assert 0.85 < float(g_c[n].norm() / (g_a[n].norm() + 1e-12)) < 1.2, n
    print(f"{model}: worst gradient cosine vs fp32: library-bf16 {worst_b:.4f}, native {worst_c:.4f}")


def test_nan_in_the_input_reaches_loss_and_weights():
    """a NaN must stay visible.  One NaN pixel in the batch -> NaN conv output -> NaN BatchNorm
    statistics -> (NaN-preserving ReLU) -> NaN loss -> NaN weights after the optimizer step, for both fused
    optimizers.  (-ffast-math would allow the compiler to assume this never happens: the numerics kernels are
    built with -fno-finite-math-only, and the ReLU is written as "x <= 0 ? 0 : x".)"""
    for opt in ("sgd", "adam"):
        eng = _engine(model="resnet18", num_classes=10, optimizer=opt, lr=0.01, sync_bn=False)
        g = torch.Generator().manual_seed(0)
        x = torch.rand(8, 3, 32, 32, generator=g).cuda()
        y = torch.randint(0, 10, (8,), generator=g).cuda()
        assert float(eng.train_step(x, y)) == float(eng.train_step(x, y)) or True  # two clean steps first
        before = [p.detach().clone() for p in eng.net.parameters()]
        assert all(torch.isfinite(p).all() for p in before)
        x[3, 1, 7, 9] = float("nan")
        loss = eng.train_step(x, y)
        assert torch.isnan(loss).item(), float(loss)
        params = dict(eng.net.module.named_parameters())
        assert torch.isnan(params["conv1.weight"]).any() and torch.isnan(params["fc.weight"]).any()
        assert torch.isnan(params["layer1.0.bn1.weight"]).any()
        lp = getattr(params["conv1.weight"], "_lp", None)
        assert lp is None or torch.isnan(lp.float()).any()  # the bf16 working copy carries it too


@pytest.mark.parametrize("shape", [(16, 64, 14, 14), (4, 256, 28, 28)])
def test_batchnorm_statistics_with_mean_100_std(shape):
    """|mean| = 100 * std: E[x^2] - E[x]^2 in fp32 over the whole tensor loses the variance; the finalize kernel
    sums the per-tile partial rows and forms the variance in double.  Compared with the fp32/fp64 oracle on the same
    bf16 tensor (torch's Welford-based batch_norm is the reference behaviour, /root/reference/main.py:82)."""
    from pytorch_distributed_training_amd.ops import native

    B, C, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    std = torch.rand(C, device="cuda", generator=g) + 0.5
    x = (torch.randn(B, C, H, W, device="cuda", generator=g) * std.view(1, C, 1, 1) + 100.0 * std.view(1, C, 1, 1))
    x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xd = x.double()
    mean_ref = xd.mean((0, 2, 3))
    var_ref = xd.var((0, 2, 3), unbiased=False)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    mean, invstd = native.bn_finalize(native.bn_stats(x), float(B * H * W), 1e-5, rm, rv, 0.1, None)
    assert ((mean.double() - mean_ref).abs() / mean_ref.abs()).max() < 1e-6
    inv_ref = 1.0 / torch.sqrt(var_ref + 1e-5)
    assert ((invstd.double() - inv_ref).abs() / inv_ref).max() < 2e-2, ((invstd.double() - inv_ref).abs() / inv_ref).max()
    # the same through the conv epilogue's per-tile statistics (identity 1x1 convolution)
    w = torch.eye(C, device="cuda").view(C, C, 1, 1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, st = native.conv2d_fwd(x, w, 1, 0, True)
    assert torch.equal(y, x)
    mean2, invstd2 = native.bn_finalize(st, float(B * H * W), 1e-5)
    assert ((invstd2.double() - inv_ref).abs() / inv_ref).max() < 2e-2


@pytest.mark.parametrize("shape", [(4, 256, 56, 56), (3, 512, 7, 9), (2, 64, 5, 5), (64, 1024, 14, 14)])
def test_bn_backward_of_two_norms_sharing_a_gradient(shape):
    """bn_bwd_elemt2 (bottleneck bn3 + downsample BN, one pass over dz and the ReLU mask) against two bn_bwd_elemt
    launches and against the fp32 formula."""
    from pytorch_distributed_training_amd.ops import native, reference

    torch.manual_seed(3)
    B, C, H, W = shape
    mk = lambda: torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dz, xa, xb = mk(), mk() * 2 + 1, mk() * 0.5 - 2
    count = float(B * H * W)
    sides = []
    for x in (xa, xb):
        mean, invstd = native.bn_finalize(native.bn_stats(x), count, 1e-5)
        sides.append((x, mean, invstd, torch.rand(C, device="cuda") + 0.5))
    z, mask = native.bn_apply_act(xa, sides[0][1], sides[0][2], sides[0][3], torch.zeros(C, device="cuda"), xb, True)
    assert mask.dtype == torch.uint8
    ps = [native.bn_bwd_reduce(dz, mask, x, m, i, True) for x, m, i, _ in sides]
    one = [native.bn_bwd_elemt(dz, mask, x, m, i, g, p, count, True, False) for (x, m, i, g), p in zip(sides, ps)]
    (xa_, ma, ia, ga), (xb_, mb_, ib, gb) = sides
    dxa, dxb, sa, sb = native.bn_bwd_elemt2(dz, mask, xa_, ma, ia, ga, ps[0], xb_, mb_, ib, gb, ps[1], count)
    for got, (want, _, sums), s2 in ((dxa, one[0], sa), (dxb, one[1], sb)):
        assert torch.allclose(s2, sums, rtol=1e-5, atol=1e-4)  # (the partial rows are summed in another order)
        assert (got.float() - want.float()).abs().max() <= 2e-2 * want.float().abs().max()
    for got, (x, m, i, g), p in ((dxa, sides[0], ps[0]), (dxb, sides[1], ps[1])):  # fp32 formula, gate = (z > 0)
        want, _, _ = reference.bn_bwd_elemt(dz.float(), z.float(), x.float(), m, i, g, p, count, True, False)
        assert (got.float() - want).abs().max() <= 2e-2 * want.abs().max()


@pytest.mark.parametrize("shape", [(4, 64, 256, 14, 14), (2, 128, 512, 9, 11), (3, 256, 1024, 7, 7), (17, 64, 256, 5, 5)])
def test_dgrad_epilogue_takes_the_sums_of_both_norms_fed_by_its_result(shape):
    """The pointwise data gradient on the GEMM core with a gated addend whose result is the gradient of a bottleneck's
    bn3 AND of the downsample BN (k_gemm_nt<..., 2, false, DUAL>): the epilogue's second pair of sums against
    bn_bwd_reduce of the same dx with the second BN's operands; dx and the first pair against the single-BN launch."""
    from pytorch_distributed_training_amd.ops import native

    torch.manual_seed(6)
    B, N, C, H, W = shape  # dy [B, N, H, W] -> dx [B, C, H, W]
    mk = lambda c: torch.randn(B, c, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy, add, x1, x2 = mk(N), mk(C), mk(C) * 2 + 1, mk(C) * 0.5 - 1
    w = (torch.randn(N, C, 1, 1, device="cuda") / C ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    count = float(B * H * W)
    m1, i1 = native.bn_finalize(native.bn_stats(x1), count, 1e-5)
    m2, i2 = native.bn_finalize(native.bn_stats(x2), count, 1e-5)
    g = torch.rand(C, device="cuda") + 0.5
    z, mask = native.bn_apply_act(x1, m1, i1, g, torch.zeros(C, device="cuda"), x2, True)
    gate = native.bn_apply_act(mk(C), m1, i1, g, torch.zeros(C, device="cuda"), None, True)[1]  # the addend's own gate
    core = 100  # (kCfgGemmCore: these shapes are small on purpose)
    dx1, ps1 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, 0, add, (x1, mask, m1, i1), gate, cfg=core)
    dx2, ps1b, ps2 = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, 0, add, (x1, mask, m1, i1), gate, cfg=core,
                                         bn_sink2=(x2, m2, i2))
    assert ps2 is not None and torch.equal(dx1, dx2)
    assert torch.allclose(ps1.sum(0), ps1b.sum(0), rtol=1e-5, atol=1e-3)
    want2 = native.bn_bwd_reduce(dx2, mask, x2, m2, i2, True).sum(0)
    got2 = ps2.sum(0)
    scale = want2.abs().max().clamp_min(1.0)
    assert ((got2 - want2).abs().max() / scale) < 2e-3, ((got2 - want2).abs().max(), scale)
    assert torch.allclose(got2[0], ps1.sum(0)[0], rtol=1e-5, atol=1e-3)  # sum g is shared


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (3, 64, 112, 112), (5, 128, 8, 12), (70, 64, 32, 32)])
def test_stem_tail_backward_gathers_the_pooled_gradient(shape):
    """bn_relu_maxpool_bwd (both BN-backward passes gather the max-pool gradient themselves; no full-resolution dz)
    against the three unfused kernels: pool backward -> bn_bwd_reduce -> bn_bwd_elemt."""
    from pytorch_distributed_training_amd.ops import native

    torch.manual_seed(4)
    B, C, H, W = shape
    u = (torch.randn(B, C, H, W, device="cuda") * 1.5 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    count = float(B * H * W)
    mean, invstd = native.bn_finalize(native.bn_stats(u), count, 1e-5)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    y, arg, mask = native.bn_relu_maxpool_fwd(u, mean, invstd, gamma, beta)
    dy = torch.randn_like(y)
    dz = native.maxpool3x3s2_bwd(dy, u, y, arg)
    ps = native.bn_bwd_reduce(dz, mask, u, mean, invstd, True)
    du0, _, sums0 = native.bn_bwd_elemt(dz, mask, u, mean, invstd, gamma, ps, count, True, False)
    du1, sums1 = native.bn_relu_maxpool_bwd(dy, arg, u, mask, mean, invstd, gamma, count)
    assert torch.allclose(sums1, sums0, rtol=1e-4, atol=1e-3 * float(sums0.abs().max()))
    assert (du1.float() - du0.float()).abs().max() <= 1e-2 * du0.float().abs().max()


def test_dataflow_fusions_do_not_change_the_native_result(monkeypatch):
    """Gated residual gradient, deferred downsample BN, compact downsample gradient, fused stem tail and the
    BN-reduce-in-dgrad epilogue only re-arrange WHERE things are computed.  Switching each off on the native path:
    the forward is bit-identical for all but the deferred downsample BN (which adds its branch in fp32 instead of
    as a bf16-rounded tensor), and the gradients agree up to the rounding of the intermediates that the fusion
    keeps in fp32.  A 14-layer bottleneck net is used: the 50-layer one at random init amplifies ONE bf16 rounding
    difference (or the run-to-run order of MIOpen's atomics) into ~10 % of the logits
    (bench/micro/fusion_ab.py prints both)."""
    import pytorch_distributed_training_amd.models.resnet as R
    import pytorch_distributed_training_amd.nn.functional as F
    from pytorch_distributed_training_amd import nn as pnn
    from pytorch_distributed_training_amd.parallel import ParamArena

    torch.manual_seed(0)
    net = R.ResNet(R.Bottleneck, [1, 1, 1, 1], num_classes=10).cuda()
    arena = ParamArena(net)
    arena.attach_lowp(torch.bfloat16)
    x = torch.randn(32, 3, 64, 64, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 10, (32,), device="cuda")
    crit = pnn.CrossEntropyLoss()
    knobs = [(R, "_LAZY_RES"), (R, "_DEFER_DS"), (R, "_COMPACT_DS"), (F, "_FUSE_STEM_POOL"), (F, "_FUSE_BN_REDUCE"),
             (F, "DUAL_BN_BACKWARD"), (F, "_FUSE_STEM_POOL_BWD")]

    def run(off=()):
        for mod, n in knobs:
            monkeypatch.setattr(mod, n, n not in off)
        for p in net.parameters():
            p.grad = None
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
        out = net(x)
        crit(out, y).backward()
        return out.detach().float(), {n: p.grad.detach().float().clone() for n, p in net.named_parameters()}

    def compare(a, b):
        (o1, g1), (o0, g0) = a, b
        err = float((o1 - o0).abs().max() / o0.abs().max())
        cos = min(float(torch.dot(g1[n].flatten(), g0[n].flatten()) / (g1[n].norm() * g0[n].norm() + 1e-12)) for n in g0)
        return err, cos

    base = run()
    assert not F.fallback_counts, dict(F.fallback_counts)
    assert compare(run(), base) == (0.0, pytest.approx(1.0, abs=1e-5))  # the native path is deterministic
    for _, name in knobs:
        err, cos = compare(run((name,)), base)
        print(f"{name} off: logits rel err {err:.4f}, worst gradient cosine {cos:.4f}")
        if name == "_DEFER_DS":
            assert err < 0.05 and cos > 0.9, (name, err, cos)
        else:
            assert err == 0.0 and cos > 0.99, (name, err, cos)


def test_toy_transformer_widths_stay_on_the_native_gemm():
    """``vit_tiny_test`` (hidden 32, MLP 64: reductions shorter than one K-tile, with residual / GELU epilogues) once
    aborted in the GEMM launcher ("unsupported epilogue combination", found by bench/cli_matrix.sh): the ragged-K
    build has no residual / GELU epilogue, so such operands are zero-padded to the K-tile."""
    from pytorch_distributed_training_amd.nn import functional as pF
    from pytorch_distributed_training_amd.ops import native

    a = (torch.randn(100, 40, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(48, 40, device="cuda") * 0.2).to(torch.bfloat16)
    res = torch.randn(100, 48, device="cuda").to(torch.bfloat16)
    bias = torch.randn(48, device="cuda")
    want = a.float() @ w.float().t() + bias + res.float()
    got = native.gemm_nt(a, w, bias, res)[0]
    assert (got.float() - want).abs().max() < 0.06
    h, u, _ = native.gemm_nt(a, w, bias, None, None, 1, True)
    assert (u.float() - (want - res.float())).abs().max() < 0.06
    assert (h.float() - torch.nn.functional.gelu(u.float())).abs().max() < 0.03
    before = dict(pF.fallback_counts)
    eng = _engine(model="vit_tiny_test", num_classes=10, optimizer="adamw", lr=1e-2, sync_bn=False)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(16, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (16,), generator=g).cuda()
    losses = [float(eng.train_step(x, y)) for _ in range(10)]
    assert all(l == l for l in losses) and losses[-1] < losses[0] - 0.3, losses
    new = {k: v - before.get(k, 0) for k, v in pF.fallback_counts.items() if v != before.get(k, 0)}
    assert not new, new  # (one head of 32, width 32: attention, LayerNorm and every GEMM on the HIP kernels)


_conv_port = [29720]


def _convergence_port():
    """(the session's own process group holds MASTER_PORT: every child job gets a port of its own)"""
    _conv_port[0] += 1
    return _conv_port[0]


def _convergence_run(backend, extra_env, extra_args, td):
    args = ["--model", "resnet18", "--image_size", "64", "--num_classes", "100", "--batch_size", "64", "--steps", "80",
            "--window", "10", "--eval_batches", "8", "--noise", "0.4", "--seed", "0"]
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "bench", "convergence.py"), "--backend", backend] + args +
                       extra_args, env={**os.environ, **extra_env, "PDT_QUIET": "1", "MASTER_ADDR": "127.0.0.1",
                                        "MASTER_PORT": str(_convergence_port()), "RANK": "0", "WORLD_SIZE": "1",
                                        "LOCAL_RANK": "0"}, cwd=td, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_native_bf16_training_follows_the_fp32_oracle_loss_curve(tmp_path):
    """Whole-system learning check the driver sees (bench/convergence.py, short form): ResNet-18 + SyncBN modules, Adam
    lr 1e-3 (the reference's optimizer, /root/reference/main.py:79-80,100-104) on learnable synthetic images, the same
    initial weights and sample stream through (a) the HIP kernels in bf16, (b) the fp32 PyTorch oracle ops, (c) the HIP
    kernels replayed from a hipGraph with bf16 gradient buckets.  Every window of 10 steps of (a) and (c) must sit
    within 0.05 + 15 % of the oracle's window (three seeds of this configuration differ by more than that among
    themselves: profiles/r3_convergence_short.md), the loss must fall by 10x, and held-out accuracy must match."""
    nat = _convergence_run("native", {}, [], tmp_path)
    ora = _convergence_run("oracle", {"PDT_FORCE_REFERENCE_OPS": "1"}, [], tmp_path)
    gra = _convergence_run("native", {}, ["--graph", "1", "--grad_dtype", "bf16"], tmp_path)
    assert nat["dtype"] == "bf16" and ora["dtype"] == "fp32" and gra["graph"] is True, (nat, ora, gra)
    for run in (nat, gra):
        assert run["finite"] and len(run["loss_curve"]) == len(ora["loss_curve"]) == 8, run
        for a, b in zip(run["loss_curve"], ora["loss_curve"]):
            assert abs(a - b) <= 0.05 + 0.15 * b, (run["loss_curve"], ora["loss_curve"])
        assert run["loss_curve"][-1] < 0.1 * run["loss_curve"][0], run
        assert run["eval_top1"] >= ora["eval_top1"] - 0.03 and run["eval_top1"] > 0.9, (run, ora)


@pytest.mark.parametrize("world,port", [(2, "29651"), (3, "29652")])
def test_two_shot_direct_allreduce_over_ipc_windows(tmp_path, world, port):
    """SURVEY 5.8 layer 4: the gradient-bucket all-reduce that uses every xGMI link at once (scatter shards into the
    peers' windows, reduce in rank order, gather).  W processes share GPU 0 (the stores stay on the device instead of
    crossing links; the protocol is the same): fp32 and bf16 buffers from 2 elements to 4 MiB against a CPU all-reduce,
    identical bits on all ranks, exact rank-order sums, the start-up self-check, hipGraph replay, oversize refusal.
    W = 3: shards that do not divide the buffer."""
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "ipc_bucket_worker.py")], env=env,
                                      cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=200)
            assert p.returncode == 0, se[-3000:]
            outs.append(json.loads(so.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for o in outs:
        assert o["setup_error"] is None, o
        assert o["worst_rel_err"]["fp32"] < 1e-6 and o["worst_rel_err"]["bf16"] < 1e-2 and o["same_bits"], o
        assert o["rank_order_exact"], o
        assert o["validate_error"] is None, o
        assert all(v == [o["graph_want"]] * 4 for v in o["graph_vals"]), o
        assert o["refused_oversize"], o


def _run_w2_pair(tmp_path, port, extra_env):
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PDT_TEST_DEADLINE_S="100", PDT_QUIET="1", **extra_env)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "bench", "micro", "engine_w2_worker.py")],
                                      env=env, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=150)
            assert p.returncode == 0, se[-3000:]
            outs.append(json.loads(so.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return sorted(outs, key=lambda o: o["rank"])


def test_two_ranks_captured_step_with_peer_exchanges_inside_equals_eager(tmp_path):
    """A hipGraph of the WHOLE step at W = 2: the SyncBN exchanges inside the BN kernels and the gradient buckets through
    the two-shot IPC all-reduce are all inside the captured launch sequence (no host-side collective in the step), on
    two processes sharing GPU 0.  The replayed run must give the eager run's losses on every rank, step by step, and
    bit-identical replicas (the ADVICE r2 condition for capturing at world size > 1: a captured step with at least two
    ranks verified against eager)."""
    common = {"PDT_SYNCBN_COMM": "ipc", "PDT_GRAD_COMM": "ipc", "PDT_TEST_STEPS": "8"}
    eager = _run_w2_pair(tmp_path, "29645", {**common, "PDT_TEST_GRAPH": "0"})
    graph = _run_w2_pair(tmp_path, "29646", {**common, "PDT_TEST_GRAPH": "1"})
    assert all(o["hip_graph"] and o["graph_failed"] is None for o in graph), graph
    assert not any(o["hip_graph"] for o in eager), eager
    for e, g in zip(eager, graph):
        assert g["losses"] == pytest.approx(e["losses"], rel=1e-5, abs=1e-6), (e["losses"], g["losses"])
        assert g["ipc_bucket_launches"] >= 5 and g["sync_bn_comm"]["path"] == "ipc", g
    assert graph[0]["bn_mean0"] == graph[1]["bn_mean0"] == eager[0]["bn_mean0"], (graph, eager)


@pytest.mark.parametrize("syncbn_comm,grad_comm,port", [("auto", "rccl", "29641"), ("ipc", "rccl", "29642"),
                                                        ("ipc", "ipc", "29643")])
def test_two_engine_ranks_on_one_gpu_train_to_identical_replicas(tmp_path, syncbn_comm, grad_comm, port):
    """W = 2 with a GPU under every rank, as far as a one-GPU box allows: two engine processes share GPU 0 behind a
    gloo group (RCCL refuses two ranks on one device).  The full step runs: HIP kernels, the C++ reducer with several
    buckets in flight (rebuilt after step 1), SyncBatchNorm statistics through the group ("auto": host-staged, the path
    a gloo job takes) or inside the BN kernels over IPC mailboxes ("ipc": the path an RCCL job on one host takes),
    gradient buckets through the host-staged group ("rccl" here means "the group") or through the two-shot direct
    all-reduce over IPC windows, in the compute stream ("ipc"), reduce_loss, the replica check.  This configuration used to stall in step 2 (profiles/r3_w2_hang.md); each rank
    has a deadline and reports where it stands if it comes back."""
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PDT_TEST_DEADLINE_S="100", PDT_SYNCBN_COMM=syncbn_comm, PDT_QUIET="1",
                   PDT_GRAD_COMM=grad_comm)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "bench", "micro", "engine_w2_worker.py")],
                                      env=env, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=150)
            assert p.returncode == 0, se[-3000:]
            outs.append(json.loads(so.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    outs.sort(key=lambda o: o["rank"])
    assert [o["world"] for o in outs] == [2, 2] and all(o["reducer"] == "native" for o in outs), outs
    assert all(o["sync_bn_comm"]["path"] == ("ipc" if syncbn_comm == "ipc" else "group") for o in outs), outs
    if grad_comm == "ipc":  # every bucket of every step went through the windows
        assert all(o["grad_comm"]["path"] == "ipc" and o["ipc_bucket_launches"] >= 5 for o in outs), outs
    else:
        assert all(o["ipc_bucket_launches"] == 0 for o in outs), outs
    l0 = outs[0]["losses"]  # rank 0 holds the mean over the ranks (reduce_loss)
    assert all(v == v for v in l0) and l0[-1] < 0.5 * l0[0], l0
    assert outs[0]["bn_mean0"] == outs[1]["bn_mean0"], outs  # running statistics: same bits on both ranks


@pytest.mark.parametrize("grad_comm", ["rccl", "ipc"])
def test_two_vit_ranks_on_one_gpu_train_to_identical_replicas(tmp_path, grad_comm):
    """The same W = 2 rehearsal with a transformer under the reducer (ViT-Ti/16 at 224 x 224: 150 parameter tensors,
    AdamW, the persistent attention kernels of both ranks sharing the CUs): gradient buckets through the host-staged
    group or through the two-shot direct all-reduce over IPC windows; the loss falls and the replicas hold the same bits."""
    procs = []
    port = str(29660 + (1 if grad_comm == "ipc" else 0))
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PDT_TEST_DEADLINE_S="100", PDT_QUIET="1", PDT_GRAD_COMM=grad_comm,
                   PDT_TEST_MODEL="vit_ti_16", PDT_TEST_STEPS="6")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "bench", "micro", "engine_w2_worker.py")],
                                      env=env, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=150)
            assert p.returncode == 0, se[-3000:]
            outs.append(json.loads(so.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    outs.sort(key=lambda o: o["rank"])
    assert [o["world"] for o in outs] == [2, 2] and all(o["reducer"] == "native" for o in outs), outs
    if grad_comm == "ipc":
        assert all(o["grad_comm"]["path"] == "ipc" and o["ipc_bucket_launches"] >= 6 for o in outs), outs
    l0 = outs[0]["losses"]
    assert all(v == v for v in l0) and l0[-1] < l0[0], l0
    assert outs[0]["param_sum"] == outs[1]["param_sum"], outs


def test_loader_sends_uint8_batches_and_the_cast_kernel_matches_the_fp32_path():
    """The loader's GPU path: uint8 NHWC through the pinned ring and the copy stream, /255 + bf16 + layout in the
    engine's one cast kernel; same tensor as the fp32 NCHW path gives, same order, short last batch."""
    from pytorch_distributed_training_amd.data import DataLoader
    from pytorch_distributed_training_amd.data.datasets import SyntheticImageNet

    ds = SyntheticImageNet(train=True, num_classes=10, length=20, pool=20, image_size=64)
    eng = _engine(model="resnet18", num_classes=10, optimizer="sgd", lr=0.01)
    dev = torch.device("cuda", torch.cuda.current_device())
    fast = DataLoader(ds, batch_size=8, pin_memory=True, device=dev)
    slow = DataLoader(ds, batch_size=8, pin_memory=True, device=dev, raw_u8=False)
    assert fast.raw_u8 and not slow.raw_u8
    n = 0
    for (xu, yu), (xf, yf) in zip(fast, slow):
        assert xu.dtype == torch.uint8 and xu.is_cuda and tuple(xu.shape[1:]) == (64, 64, 3)
        assert xf.dtype == torch.float32 and tuple(xf.shape[1:]) == (3, 64, 64)
        a, b = eng.prepare_input(xu), eng.prepare_input(xf)
        assert a.dtype == torch.bfloat16 and a.is_contiguous(memory_format=torch.channels_last) and a.shape == b.shape
        assert (a.float() - b.float()).abs().max() <= 2 ** -8  # (x * (1/255) vs x / 255 before the bf16 rounding)
        assert torch.equal(yu, yf)
        n += xu.shape[0]
    assert n == 20
    x, y = next(iter(fast))
    loss = float(eng.train_step(x, y))  # the uint8 batch goes through the step as it is
    assert loss == loss and loss < 20.0


def test_no_sync_accumulates_gradients_on_the_native_path():
    """Gradient accumulation (``DistributedDataParallel.no_sync``) with bucket-view gradients on the GPU: a backward
    inside no_sync followed by a synchronised one leaves grad(x1) + grad(x2) in ``.grad``."""
    eng = _engine(model="resnet18", num_classes=10, optimizer="sgd", lr=0.0)
    net, crit = eng.net, eng.criterion
    g = torch.Generator().manual_seed(0)
    xs = [torch.rand(16, 3, 64, 64, generator=g).cuda() for _ in range(2)]
    ys = [torch.randint(0, 10, (16,), generator=g).cuda() for _ in range(2)]

    def fb(i):
        crit(net(eng.prepare_input(xs[i])), ys[i]).backward()

    def grads():
        return [p.grad.detach().float().clone() for p in net.parameters()]

    fb(0)  # warm-up iteration: the reducer rebuilds its buckets after the first synchronised backward
    eng.optimizer.zero_grad()
    fb(0)
    g1 = grads()
    eng.optimizer.zero_grad()
    fb(1)
    g2 = grads()
    eng.optimizer.zero_grad()
    with net.no_sync():
        fb(0)
    fb(1)
    acc = grads()
    buckets = {t.untyped_storage().data_ptr() for t in net.reducer.bucket_tensors()}
    assert all(p.grad.untyped_storage().data_ptr() in buckets for p in net.parameters())
    for a, b, c, (n, _) in zip(acc, g1, g2, net.module.named_parameters()):
        want = b + c
        assert (a - want).abs().max() <= 1e-3 * float(want.abs().max()) + 1e-6, n


@pytest.mark.parametrize("model,size", [("resnet50", 96), ("resnet18", 64)])
def test_eval_mode_on_the_native_path_matches_the_oracle(model, size):
    """Inference (``net.eval()``: BN with running statistics, no autograd) through the HIP kernels against the same
    network run through the fp32 oracle ops, after a few training steps have moved the running statistics."""
    from pytorch_distributed_training_amd import ops

    eng = _engine(model=model, num_classes=10, optimizer="sgd", lr=0.01)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(16, 3, size, size, generator=g).cuda()
    y = torch.randint(0, 10, (16,), generator=g).cuda()
    for _ in range(3):
        eng.train_step(x, y)
    rm = eng.net.module.bn1.running_mean.float()
    assert float(rm.abs().sum()) > 0
    eng.net.eval()
    try:
        with torch.no_grad():
            xb = eng.prepare_input(x)
            out = eng.net(xb).float()
            ops.set_force_reference(True)
            try:
                ref = eng.net(xb).float()
            finally:
                ops.set_force_reference(False)
    finally:
        eng.net.train()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max() < 0.05 * max(1.0, float(ref.abs().max())), float((out - ref).abs().max())
    assert torch.equal(eng.net.module.bn1.running_mean.float(), rm)  # eval does not touch the running statistics
    acc = eng.evaluate([(x, y)])
    assert 0.0 <= acc <= 1.0


def test_label_outside_the_head_makes_the_loss_nan():
    """ATen raises a device-side assertion for a label >= classes; the fused loss kernel reads nothing out of
    bounds and returns NaN, so that a head narrower than the dataset's label range cannot train on garbage."""
    from pytorch_distributed_training_amd.ops import native

    logits = torch.randn(8, 16, device="cuda").to(torch.bfloat16)
    labels = torch.randint(0, 16, (8,), device="cuda")
    loss, _ = native.softmax_ce_fwd_bwd(logits, labels)
    assert float(loss) == float(loss)
    labels[3] = 16
    loss, dl = native.softmax_ce_fwd_bwd(logits, labels)
    assert float(loss) != float(loss) and bool(torch.isfinite(dl.float()).all())


@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_checkpoint_resume_on_the_native_path_is_bit_exact(tmp_path, opt):
    """SURVEY 5.4 on the GPU path: fp32 master weights in the arena, their bf16 shadows, BN running statistics and
    the optimizer state go through a checkpoint; a fresh engine (other seed) that loads it continues with the losses
    of the uninterrupted run, bit for bit (the kernels are deterministic)."""
    from pytorch_distributed_training_amd.utils import checkpoint as ckpt

    g = torch.Generator().manual_seed(0)
    xs = [torch.rand(16, 3, 64, 64, generator=g).cuda() for _ in range(5)]
    ys = [torch.randint(0, 10, (16,), generator=g).cuda() for _ in range(5)]
    kw = dict(model="resnet18", num_classes=10, optimizer=opt, lr=0.01 if opt == "sgd" else 1e-3)
    a = _engine(**kw)
    first = [float(a.train_step(xs[i], ys[i])) for i in range(3)]
    path = str(tmp_path / "ck.pt")
    assert ckpt.save_checkpoint(path, a.net, a.optimizer, epoch=0, global_step=3)
    rest = [float(a.train_step(xs[i], ys[i])) for i in range(3, 5)]
    b = _engine(seed=123, **kw)
    w0 = b.net.module.layer1[0].conv1.weight
    assert not torch.equal(w0.detach(), a.net.module.layer1[0].conv1.weight.detach())
    st = ckpt.load_checkpoint(path, b.net, b.optimizer)
    assert st["global_step"] == 3
    assert torch.equal(w0._lp, w0.detach().to(torch.bfloat16))  # shadows refreshed from the loaded masters
    resumed = [float(b.train_step(xs[i], ys[i])) for i in range(3, 5)]
    assert resumed == rest, (first, rest, resumed)
    for (n, p), q in zip(a.net.module.state_dict().items(), b.net.module.state_dict().values()):
        assert torch.equal(p, q), n


def test_plain_torch_calibration_mode_trains_every_parameter(tmp_path):
    """PDT_PLAIN_TORCH=1 (the same-hardware calibration row: stock MIOpen / ATen ops) must run the WHOLE training
    step: every parameter, the convolution weights included, gets a gradient and moves.  (It once read the detached
    bf16 shadow of the weights: no weight gradient, a calibration row that skipped a third of the work and a
    network that did not learn -- bench/convergence.py.)"""
    code = (
        "import os, sys, json, torch\n"
        "sys.path.insert(0, %r)\n"
        "from pytorch_distributed_training_amd import dist as pdist\n"
        "from pytorch_distributed_training_amd.engine import Engine, EngineConfig\n"
        "pdist.init_process_group(backend='rccl', local_rank=0)\n"
        "eng = Engine(EngineConfig(model='resnet18', num_classes=10, dtype='bf16', optimizer='sgd', lr=0.05, seed=0))\n"
        "g = torch.Generator().manual_seed(0)\n"
        "x = torch.rand(32, 3, 64, 64, generator=g).cuda(); y = torch.randint(0, 10, (32,), generator=g).cuda()\n"
        "before = {n: p.detach().clone() for n, p in eng.net.module.named_parameters()}\n"
        "losses = [float(eng.train_step(x, y)) for _ in range(6)]\n"
        "nograd = [n for n, p in eng.net.module.named_parameters() if p.grad is None or float(p.grad.abs().sum()) == 0.0]\n"
        "still = [n for n, p in eng.net.module.named_parameters() if torch.equal(p.detach(), before[n])]\n"
        "print(json.dumps({'losses': losses, 'nograd': nograd, 'still': still}))\n"
    ) % ROOT
    env = dict(os.environ, PDT_PLAIN_TORCH="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29553", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["nograd"] == [] and out["still"] == [], out
    assert out["losses"][-1] < out["losses"][0] - 0.3, out


def test_reducer_runs_the_rccl_path_with_a_size1_communicator(tmp_path):
    """The multi-GPU hot path (multi-tensor flatten -> fence -> ncclAllReduce on the communicator's stream ->
    per-bucket event -> optimizer) on one GPU: PDT_FORCE_COMM=1 gives the C++ reducer a size-1 communicator, and
    the run must reproduce the losses of the plain W=1 run bit for bit."""
    code = (
        "import os, sys, json, torch\n"
        "sys.path.insert(0, %r)\n"
        "from pytorch_distributed_training_amd import dist as pdist\n"
        "from pytorch_distributed_training_amd.engine import Engine, EngineConfig\n"
        "pdist.init_process_group(backend='rccl', local_rank=0)\n"
        "eng = Engine(EngineConfig(model='resnet18', num_classes=10, dtype='bf16', optimizer='sgd', lr=0.05, seed=0))\n"
        "g = torch.Generator().manual_seed(0)\n"
        "x = torch.rand(32, 3, 64, 64, generator=g).cuda(); y = torch.randint(0, 10, (32,), generator=g).cuda()\n"
        "losses = [float(eng.train_step(x, y)) for _ in range(6)]\n"
        "stats = None\n"
        "if eng.net.reducer.set_timing(True):\n"
        "    losses.append(float(eng.train_step(x, y)))\n"
        "    stats = eng.net.reducer.comm_stats()\n"
        "    eng.net.reducer.set_timing(False)\n"
        "else:\n"
        "    losses.append(float(eng.train_step(x, y)))\n"
        "print(json.dumps({'losses': losses, 'rccl': eng.net.reducer.rccl is not None, 'kind': eng.net.reducer_kind, 'comm': stats}))\n"
    ) % ROOT
    outs = []
    for force in ("0", "1"):
        env = dict(os.environ, PDT_FORCE_COMM=force, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", RANK="0",
                   WORLD_SIZE="1", LOCAL_RANK="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=tmp_path)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["kind"] == outs[1]["kind"] == "native"
    assert not outs[0]["rccl"] and outs[1]["rccl"]
    assert outs[0]["losses"] == outs[1]["losses"], outs
    assert outs[1]["losses"][-1] < outs[1]["losses"][0]
    # per-bucket all-reduce timing (the JSON summary's "comm" block): only with the communicator
    c = outs[1]["comm"]
    assert outs[0]["comm"] is None and c is not None
    assert len(c["bucket_allreduce_ms"]) == len(c["bucket_bytes"]) >= 2 and all(v >= 0 for v in c["bucket_allreduce_ms"])
    assert 0.0 <= c["overlap_fraction"] <= 1.0 and c["allreduce_span_ms"] >= 0 and c["exposed_ms"] >= 0


def test_hip_graph_with_rccl_allreduce_debug_and_shutdown(tmp_path):
    """Multi-rank hipGraph rehearsal on one GPU: with a size-1 communicator (PDT_FORCE_COMM=1) the captured step
    contains the flatten -> fence -> ncclAllReduce on the communicator's stream -> event join, and its replays must
    give the eager losses bit for bit.  Runs with the reducer's debug assertions and roctx ranges on, and ends with
    the ordered teardown (watchdog join -> stream drain -> ncclCommDestroy -> c10d)."""
    code = (
        "import os, sys, json, torch\n"
        "sys.path.insert(0, %r)\n"
        "from pytorch_distributed_training_amd import dist as pdist\n"
        "from pytorch_distributed_training_amd.utils import profiling\n"
        "from pytorch_distributed_training_amd.engine import Engine, EngineConfig\n"
        "pdist.init_process_group(backend='rccl', local_rank=0)\n"
        "profiling.enable_roctx(True)\n"
        "use_graph = os.environ['USE_GRAPH'] == '1'\n"
        "eng = Engine(EngineConfig(model='resnet18', num_classes=10, dtype='bf16', optimizer='sgd', lr=0.05, seed=0,\n"
        "                          graph=use_graph, check_replicas=True, check_replicas_every=3))\n"
        "g = torch.Generator().manual_seed(0)\n"
        "x = torch.rand(32, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 10, (32,), generator=g).cuda()\n"
        "losses = [float(eng.train_step(x, y)) for _ in range(9)]\n"
        "out = {'losses': losses, 'rccl': eng.net.reducer.rccl is not None, 'graph': eng._graph is not None,\n"
        "       'failed': eng._graph_failed, 'checks': eng.net.reducer.debug_checks(), 'log': eng.net.reducer.launch_log(),\n"
        "       'nb': len(eng.net.reducer.bucket_tensors())}\n"
        "comm = eng.net.reducer.rccl\n"
        "pdist.shutdown(eng)\n"
        "out['closed'] = bool(comm.closed); out['init_after'] = pdist.is_initialized()\n"
        "print(json.dumps(out))\n"
    ) % ROOT
    outs = []
    for use_graph in ("0", "1"):
        env = dict(os.environ, PDT_FORCE_COMM="1", USE_GRAPH=use_graph, MASTER_ADDR="127.0.0.1", MASTER_PORT="29553",
                   RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=tmp_path)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    eager, graph = outs
    assert eager["rccl"] and graph["rccl"] and graph["graph"], graph
    assert eager["losses"] == graph["losses"], (eager["losses"], graph["losses"])
    assert eager["checks"] > 0 and eager["log"] == list(range(eager["nb"]))
    assert eager["closed"] and graph["closed"] and not eager["init_after"] and not graph["init_after"]


def test_ipc_small_allreduce_two_processes_on_one_gpu(tmp_path):
    """SURVEY 5.8 layer 3 / 2.5 S4: the one-kernel all-reduce over IPC-mapped mailboxes, with two processes that
    share GPU 0 (the only multi-process GPU set-up a one-GPU box allows; the peers' stores then stay on the device
    instead of crossing xGMI, the protocol is the same).  Checks sums against a CPU all-reduce, bit-identical results
    on both ranks, the path-selection self-check, replay inside a hipGraph, SyncBatchNorm through the IPC path
    against plain BN over the concatenated batch, and the bounded wait when a peer never arrives."""
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29563", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "ipc_worker.py")], env=env,
                                      cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=240)
            assert p.returncode == 0, se[-3000:]
            outs.append(json.loads(so.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for o in outs:
        assert o["setup_error"] is None, o
        assert o["worst_rel_err"] < 1e-6 and o["same_bits"], o
        assert o["validate_error"] is None, o
        assert o["graph_vals"] == [o["graph_want"]] * 3, o
        assert o["syncbn_path"] == "ipc", o
        assert o["syncbn_fwd_err"] < 0.05 and o["syncbn_dx_err"] < 0.02, o
        assert o["syncbn_rm_err"] < 1e-4 and o["syncbn_rv_err"] < 1e-3, o
        # the exchange runs inside the BN kernels: same launches as a plain BN, in both backward variants
        for tag, (k_ref, k_sync) in o["kernel_names"].items():
            assert sorted(k_ref) == sorted(k_sync) and len(k_ref) >= 4, (tag, k_ref, k_sync)
        assert o["syncbn_big_mean_err"] < 1e-4 and o["syncbn_big_var_rel_err"] < 2e-3, o  # (fp32 partial rows inside a tile)
        # bottleneck net: 4 downsample blocks, dual BN backward with in-kernel exchange in the SyncBN run only
        assert o["dual_calls"] == [4, 4, True, False], o
        assert o["dual_out_err"] < 0.05 and o["dual_dw_cos"] > 0.9, o
        # a captured step with the exchange inside its kernels replays to the eager losses on every rank
        assert o["captured_losses"][1] == pytest.approx([o["captured_losses"][0][-1]] * 3, rel=1e-5), o
        # ordered teardown: the SyncBN path's mailboxes were there, close_all() closed them and forgot them
        assert o["closed_paths"][0] >= 1 and o["closed_paths"][1] == o["closed_paths"][0], o
        assert o["closed_paths"][2] == 0 and o["closed_paths"][3] and o["red_closed"], o
    r0 = [o for o in outs if o["rank"] == 0][0]
    assert r0["timed_out"] and r0["refused_after_timeout"], r0
    # peer missing inside a replayed graph: ONE time limit (not one per exchange), then the next replay raises
    pm = r0["captured_peer_missing"]
    assert pm["raised"] and pm["limit_s"] * 0.9 < pm["elapsed_s"] < pm["limit_s"] * 1.8, pm
    print("ipc all-reduce us/op:", r0["ipc_us"], " gloo-staged group us/op:", r0["group_us"])
    print("us per BN layer (fwd+bwd), 2 ranks on one GPU:", r0["layer_us"])
    # the in-kernel exchange must not cost more than the packed tensor + all-reduce kernel path it replaces
    assert r0["layer_us"]["syncbn_inkernel"] < r0["layer_us"]["syncbn_packed"] * 1.05, r0["layer_us"]


def test_bf16_gradient_buckets_follow_the_fp32_run():
    """``grad_dtype="bf16"``: gradients are cast into bf16 buckets by the multi-tensor flatten kernel and cast back
    into the fp32 ``.grad`` by ONE unflatten launch per bucket; training follows the fp32-bucket run to bf16
    precision and the buckets really are bf16."""
    g = torch.Generator().manual_seed(0)
    x = torch.rand(32, 3, 64, 64, generator=g).cuda()
    y = torch.randint(0, 10, (32,), generator=g).cuda()
    runs = {}
    for gd in ("fp32", "bf16"):
        eng = _engine(model="resnet18", num_classes=10, optimizer="sgd", lr=0.05, sync_bn=False, grad_dtype=gd)
        runs[gd] = [float(eng.train_step(x, y)) for _ in range(6)]
        dt = {t.dtype for t in eng.net.reducer.bucket_tensors()}
        assert dt == ({torch.bfloat16} if gd == "bf16" else {torch.float32})
        assert all(p.grad.dtype == torch.float32 for p in eng.net.parameters())
    # (two training runs that differ in the rounding of every gradient: identical first loss, then the trajectories
    # separate while the loss falls 30x on the one batch; the later steps are only required to stay in the same range)
    assert runs["bf16"][:2] == pytest.approx(runs["fp32"][:2], rel=0.02), runs
    assert runs["bf16"] == pytest.approx(runs["fp32"], rel=0.25, abs=0.05), runs
    assert runs["bf16"][-1] < 0.5 * runs["bf16"][0] and runs["fp32"][-1] < 0.5 * runs["fp32"][0]


@pytest.mark.parametrize("opt,gd", [("sgd", "fp32"), ("adam", "fp32"), ("adam", "bf16")])
def test_hip_graph_step_matches_eager(opt, gd):
    """Engine(graph=True) captures forward+backward+optimizer after 3 eager steps; replays must produce the same
    losses as the eager engine (Adam's bias correction comes from the device-side step counter).  With bf16 gradient
    buckets the fp32 ``.grad`` tensors are not bucket views: the capture allocates them anew, and the optimizer's
    pointer table for them is uploaded through kernel arguments (a host-to-device copy is not capturable)."""
    g = torch.Generator().manual_seed(0)
    xs = [torch.rand(32, 3, 32, 32, generator=g).cuda() for _ in range(2)] * 4  # two alternating batches
    ys = [torch.randint(0, 10, (32,), generator=g).cuda() for _ in range(2)] * 4
    runs = []
    for graph in (False, True):
        eng = _engine(model="resnet18", num_classes=10, optimizer=opt, lr=0.01 if opt == "sgd" else 1e-3, graph=graph,
                      grad_dtype=gd)
        eng.optimizer.set_capturable(True)  # same (device-side) bias-correction arithmetic in both runs
        losses = [float(eng.train_step(x, y)) for x, y in zip(xs, ys)]
        runs.append(losses)
        if graph:
            assert eng._graph is not None, eng._graph_failed
            if opt == "adam":  # the device-side counter advanced on every replay and is what state_dict() reports
                sd = eng.optimizer.state_dict()
                assert all(int(v["step"]) == 8 for v in sd["state"].values())
    assert runs[0] == pytest.approx(runs[1], rel=2e-3), runs
    assert runs[1][-2] < runs[1][0] and runs[1][-1] < runs[1][1], runs


def test_hip_graph_vit_step_matches_eager():
    """A ViT step (197 tokens: the persistent pipelined attention kernels, LayerNorm, the GEMM core with every fused
    epilogue, AdamW) captured in one hipGraph replays to the eager losses; no library fallback in either run."""
    from pytorch_distributed_training_amd.nn import functional as pF

    g = torch.Generator().manual_seed(1)
    xs = [torch.rand(6, 3, 224, 224, generator=g).cuda() for _ in range(2)] * 4
    ys = [torch.randint(0, 10, (6,), generator=g).cuda() for _ in range(2)] * 4
    before = dict(pF.fallback_counts)
    runs = []
    for graph in (False, True):
        eng = _engine(model="vit_ti_16", num_classes=10, optimizer="adamw", lr=1e-3, graph=graph, image_size=224)
        eng.optimizer.set_capturable(True)
        runs.append([float(eng.train_step(x, y)) for x, y in zip(xs, ys)])
        if graph:
            assert eng._graph is not None, eng._graph_failed
    assert dict(pF.fallback_counts) == before, dict(pF.fallback_counts)
    assert runs[0] == pytest.approx(runs[1], rel=5e-3), runs
    assert runs[1][-2] < runs[1][0], runs


def test_batch_chunking_beyond_the_descriptor_window_matches_single_launches(kc):
    """Activations larger than one buffer descriptor (2 GiB: B > 1337 at 224x224) run as several launches over
    batch chunks.  The window is shrunk to 192 KiB so that small tensors take that path: outputs that do not
    depend on the chunking (y, dx, pooled values, masks) must be bit-identical to the single-launch run, stacked
    partial rows (BN statistics, split-K slabs) equal up to the order of fp32 additions; then a ResNet-18 step."""
    from pytorch_distributed_training_amd.ops import native
    ext = pdt.require_native()
    torch.manual_seed(3)

    def run_all():
        out = {}
        for (B, C, H, W, N, k, s, p) in [(7, 64, 14, 14, 128, 3, 1, 1), (7, 128, 14, 14, 64, 3, 2, 1),
                                          (9, 256, 8, 8, 64, 1, 1, 0), (5, 64, 12, 12, 128, 1, 2, 0)]:
            g = torch.Generator(device="cuda").manual_seed(B * C + N)
            x = torch.randn(B, C, H, W, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(N, C, k, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            y, st = native.conv2d_fwd(x, w, s, p, True)
            dy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            tag = f"{C}->{N}k{k}s{s}"
            out[tag + " y"] = y
            out[tag + " stats~"] = native.bn_sum_partials(st)
            out[tag + " dx"] = native.conv2d_dgrad(dy, w, (B, C, H, W), s, p)
            out[tag + " dw~"] = native.conv2d_wgrad(dy, x, (N, C, k, k), s, p)
            if s == 1:
                xb = (torch.randn(B, C, H, W, device="cuda", generator=g) + 0.2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                mean, invstd = native.bn_finalize(native.bn_stats(xb), float(B * H * W), 1e-5)
                one = torch.ones(C, device="cuda")
                _, act = native.bn_apply_act(xb, mean, invstd, one, one * 0.1, None, True)
                add = torch.randn(B, C, H, W, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                dxb, ps = native.conv2d_dgrad(dy, w, (B, C, H, W), 1, p, add, (xb, act, mean, invstd), act)
                out[tag + " dx_fused"] = dxb
                out[tag + " psums~"] = native.bn_sum_partials(ps)
        xs = torch.randn(14, 3, 32, 32, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ws = (torch.randn(64, 3, 7, 7, device="cuda") * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        torch.manual_seed(5)
        ys, sts, xp = ext.conv_stem_fwd(xs, ws, True)
        out["stem y"], out["stem stats~"] = ys, native.bn_sum_partials(sts)
        out["stem dw~"] = ext.conv_stem_wgrad(torch.ones_like(ys) * 0.5, xp)
        mean, invstd = native.bn_finalize(sts, float(14 * 16 * 16), 1e-5)
        one = torch.ones(64, device="cuda")
        yp, arg, mask = ext.bn_relu_maxpool_fwd(ys, mean, invstd, one, one * 0.05)
        out["pool y"], out["pool arg"], out["pool mask"] = yp, arg, mask
        out["pool dx"] = ext.maxpool3x3s2_bwd(torch.ones_like(yp), arg, list(ys.shape))
        y2, a2 = ext.maxpool3x3s2_fwd(ys)
        out["maxpool y"], out["maxpool arg"] = y2, a2
        return out

    torch.manual_seed(11)
    ref = run_all()
    ext.set_activation_window(192 * 1024)
    try:
        torch.manual_seed(11)
        got = run_all()
        torch.manual_seed(0)
        eng = _engine(model="resnet18", num_classes=10, optimizer="sgd", lr=0.01)
        g = torch.Generator().manual_seed(0)
        x = torch.rand(24, 3, 64, 64, generator=g).cuda()
        y = torch.randint(0, 10, (24,), generator=g).cuda()
        small = [float(eng.train_step(x, y)) for _ in range(3)]
    finally:
        ext.set_activation_window((1 << 31) - 4096)
    for k, v in ref.items():
        if k.endswith("~"):
            err = float((got[k].float() - v.float()).abs().max() / v.float().abs().max().clamp_min(1e-20))
            assert err < 2e-5, (k, err)
        else:
            assert torch.equal(got[k], v), k
    torch.manual_seed(0)
    eng = _engine(model="resnet18", num_classes=10, optimizer="sgd", lr=0.01)
    full = [float(eng.train_step(x, y)) for _ in range(3)]
    # same forward (first loss), then the fp32 summation order of the stacked partial rows separates the runs slowly
    assert abs(small[0] - full[0]) < 1e-3 and all(abs(a - b) < 5e-2 * max(1.0, abs(b)) for a, b in zip(small, full)), (small, full)


def test_batch_sizing_model_resnet50():
    """ResNet-50 at 224x224: ~90 MB of saved bf16 activations per image; the 288 GB of HBM bound the per-GPU batch
    (the 2 GiB buffer-descriptor range of the widest activation only decides how many launches a layer takes)."""
    from pytorch_distributed_training_amd.utils.memory import suggest_batch

    eng = _engine(optimizer="sgd", lr=0.01, sync_bn=False)
    m = suggest_batch(eng, (3, 224, 224), 1000, probe=8)
    assert 20e6 < m["bytes_per_image"] < 200e6, m
    assert m["by_memory"] > 2048 and 1024 <= m["by_tensor_range"] < 2048, m
    assert m["batch"] == m["by_memory"] // 64 * 64 and m["batch"] > m["by_tensor_range"], m


def test_main_py_single_gpu_through_launcher(tmp_path):
    """SURVEY §7.3 exit criteria: loss decreases, byte-exact log header, profiler trace, strict checkpoint."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "pytorch_distributed_training_amd.launch", "--nproc_per_node=1",
           "--master_port=29547", os.path.join(ROOT, "main.py"), "--batch_size", "64", "--dataset", "synthetic-cifar",
           "--dataset_len", "1920", "--epochs", "1", "--steps", "30", "--checkpoint", "ck.pt", "--summary_json", "s.json"]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = open(tmp_path / "Job0_64_0.log").read().splitlines()
    assert lines[0] == "datetime\tg_step\tg_img\tloss_value\texamples_per_sec"
    rows = [l.split("\t") for l in lines[1:-1]]
    assert [int(x[1]) for x in rows] == [5, 10, 15, 20, 25, 30]
    assert float(rows[-1][3]) < float(rows[0][3])
    assert list((tmp_path / "log_Job0").glob("*.pt.trace.json")), "profiler trace missing"
    s = json.load(open(tmp_path / "s.json"))
    assert s["reducer"] == "native" and s["native_extension"] and s["dtype"] == "torch.bfloat16"
    assert s["peak_hbm_bytes"] > 0 and s["images_per_sec_synchronised"] > 0
    # --graph auto: a single-process run on 32x32 images is launch-bound, the steady-state step is replayed from a
    # hipGraph captured after the profiler's cycle
    assert s["hip_graph"] and s["hip_graph_failed"] is None, s
    from pytorch_distributed_training_amd.models import resnet50

    ck = torch.load(tmp_path / "ck.pt", weights_only=True)
    resnet50().load_state_dict(ck["model"], strict=True)


def test_resnet152_bf16_step_many_buckets():
    """BASELINE.json config 4: the deep model (465 parameter tensors, 60 M parameters) steps through the native
    path and its gradients spread over many 16 MiB buckets."""
    eng = _engine(model="resnet152", optimizer="sgd", lr=0.02, sync_bn=False)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(16, 3, 64, 64, generator=g).cuda()
    y = torch.randint(0, 1000, (16,), generator=g).cuda()
    losses = [float(eng.train_step(x, y)) for _ in range(5)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    assert sum(p.numel() for p in eng.net.parameters()) == 60192808
    assert len(eng.net.reducer.bucket_tensors()) >= 12


def test_vit_b16_bf16_step():
    eng = _engine(model="vit_b_16", optimizer="adamw", lr=1e-3, sync_bn=False)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(8, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (8,), generator=g).cuda()
    losses = [float(eng.train_step(x, y)) for _ in range(6)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
