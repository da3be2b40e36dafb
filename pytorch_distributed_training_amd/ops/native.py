"""HIP-kernel implementations (gfx950) with the same signatures as ``ops.reference``.

Every function here launches hand-written kernels from the in-tree ``_C``
extension on the current HIP stream.  ``*_unsupported_reason`` functions say
why a particular operand combination cannot take the native path (the caller
then uses the library/oracle path and counts the fallback, so it shows up in
the run summary instead of silently changing what executes).
"""
from __future__ import annotations

import torch

from ..utils.env import require_native
from . import reference

_C = require_native()

_BF16 = torch.bfloat16


# ------------------------------------------------------------------ support
def _is_cl(t: torch.Tensor) -> bool:
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def conv_unsupported_reason(x, w, stride: int, padding: int):
    if x.dtype != _BF16 or w.dtype != _BF16:
        return f"dtype {x.dtype}/{w.dtype} (native conv is bf16)"
    cout, cin, kh, kw = w.shape
    if _is_stem(w, stride, padding):
        if x.shape[2] % 2 or x.shape[3] % 2 or x.shape[3] // 2 > 112 or cout % 64:
            return f"stem input {tuple(x.shape)}"
        return None
    if cin % 64 or cout % 64:
        return f"channels {cin}->{cout} not multiples of 64"
    if kh * kw > 49 or stride not in (1, 2):
        return f"kernel {kh}x{kw} stride {stride}"
    # (activations of any size: the bindings cut the batch into chunks that fit one buffer descriptor)
    if x.numel() // max(x.shape[0], 1) * 2 * max(cout // max(cin, 1), 1) >= (1 << 31):
        return "one image >= 2 GiB"
    return None


def bn_unsupported_reason(x):
    if x.dtype != _BF16:
        return f"dtype {x.dtype} (native BN/pool kernels are bf16)"
    c = x.shape[1]
    if c % 8 or c // 8 > 256 or 256 % (c // 8):
        return f"channel count {c}"
    return None


def _is_stem(w, stride, padding) -> bool:
    """7x7 / 2 / pad 3 on 3 input channels: runs on the packed image of csrc/kernels/conv_stem.hip"""
    return tuple(w.shape[1:]) == (3, 7, 7) and stride == 2 and padding == 3


def _w_cl(w):
    return w if _is_cl(w) else w.contiguous(memory_format=torch.channels_last)


# --------------------------------------------------------------------- conv
# Transposed weights ([Cin][taps][Cout]) for the data-gradient GEMM.  Parameters' bf16 working copies are
# long-lived tensors that only change in optimizer.step() / load_state_dict, so their transposes are
# cached per "weights epoch" and ALL refreshed by one multi-tensor launch at the first dgrad of a step.
_weights_epoch = 0
_wt_registry: dict[int, list] = {}  # id(w) -> [w, wt, epoch]


def bump_weights_epoch() -> None:
    """Call whenever parameter values changed (optimizer step, checkpoint load)."""
    global _weights_epoch
    _weights_epoch += 1


def _register_weight(w) -> None:
    if getattr(w, "_pdt_param", False) and id(w) not in _wt_registry:
        _wt_registry[id(w)] = [w, None, -1]


def _transposed(w):
    if not getattr(w, "_pdt_param", False):
        return _C.transpose_weight(_w_cl(w))
    ent = _wt_registry.get(id(w))
    if ent is None:
        ent = _wt_registry[id(w)] = [w, None, -1]
    if ent[2] != _weights_epoch:
        stale = [e for e in _wt_registry.values() if e[2] != _weights_epoch and e[0].device == w.device]
        have = all(e[1] is not None for e in stale)
        outs = _C.transpose_weights([e[0] for e in stale], [e[1] for e in stale] if have else None)
        for e, o in zip(stale, outs):
            e[1], e[2] = o, _weights_epoch
    return ent[1]


def conv2d_fwd(x, w, stride: int, padding: int, want_stats: bool = False, return_saved: bool = False, cfg: int = 0):
    """-> (y, stats[, saved]): ``saved`` is the tensor ``conv2d_wgrad`` wants as its ``x`` (the input
    itself, or the stem's packed image)."""
    if _is_stem(w, stride, padding):
        y, stats, saved = _C.conv_stem_fwd(x, _w_cl(w), want_stats)
    else:
        _register_weight(w)
        y, stats = _C.conv2d_fwd(x, _w_cl(w), stride, padding, want_stats, cfg)
        saved = x
    stats = stats if want_stats else None
    return (y, stats, saved) if return_saved else (y, stats)


def conv2d_dgrad(dy, w, x_shape, stride: int, padding: int, addend=None, bn_sink=None, addend_mask=None,
                 addend_stride: int = 1, cfg: int = 0, bn_sink2=None):
    """``bn_sink = (x, mask, mean, invstd)`` of the BN+ReLU that produced this conv's input: the kernel epilogue
    then also reduces that BN's backward sums and the call returns ``(dx, psums)`` (stride 1 only).
    ``bn_sink2 = (x2, mean2, invstd2)``: a second BN fed by the same gradient through the same gate (the downsample
    BN next to a bottleneck's bn3): the call returns ``(dx, psums, psums2)``, ``psums2`` None when the launch did not
    go to the kernel that takes it (the caller then reduces on its own).
    ``addend_mask``: 1-bit activity mask ``[pixels, C/8]`` gating the addend (stride 1 only)."""
    kh, kw = w.shape[2], w.shape[3]
    wt = _transposed(w)  # [Cin][taps][Cout]: K-contiguous operand for the dgrad GEMM
    if addend_stride != 1:
        assert stride == 1 and addend_mask is None, "compact stride-2 addend: stride-1 data gradients only"
    if addend_mask is not None and stride != 1:
        addend, addend_mask = apply_relu_gate(addend, addend_mask), None
    if bn_sink is not None and bn_sink2 is not None:
        return _C.conv2d_dgrad(dy, wt, list(x_shape), kh, kw, stride, padding, addend, *bn_sink, addend_mask,
                               addend_stride, cfg, *bn_sink2)
    if bn_sink is not None:
        return _C.conv2d_dgrad(dy, wt, list(x_shape), kh, kw, stride, padding, addend, *bn_sink, addend_mask,
                               addend_stride, cfg)[:2]
    if addend is not None and stride != 1:
        return _C.conv2d_dgrad(dy, wt, list(x_shape), kh, kw, stride, padding, None, cfg=cfg)[0] + addend
    return _C.conv2d_dgrad(dy, wt, list(x_shape), kh, kw, stride, padding, addend, None, None, None, None,
                           addend_mask, addend_stride, cfg)[0]


def apply_relu_gate(g, mask):
    """``g`` where the bit of the 1-bit activity mask ``[pixels, C/8]`` is set, 0 elsewhere (slow path: the fused
    kernels take the mask directly)."""
    b, c, h, w = g.shape
    bits = (mask.view(-1, c // 8, 1) >> torch.arange(8, device=mask.device, dtype=torch.uint8).view(1, 1, 8)) & 1
    gate = bits.view(b, h, w, c).permute(0, 3, 1, 2)
    return (g * gate.to(g.dtype)).contiguous(memory_format=torch.channels_last)


def conv2d_wgrad(dy, x, w_shape, stride: int, padding: int, splits: int = 0):
    if tuple(w_shape[1:]) == (3, 7, 7) and stride == 2 and padding == 3:
        if x.shape[-1] != 4:  # the plain input: pack it (callers on the autograd path hand over the saved image)
            x = _C.stem_pack_x(x)
        return _C.conv_stem_wgrad(dy, x)
    return _C.conv2d_wgrad(dy, x, w_shape[2], w_shape[3], stride, padding, splits)


# ---------------------------------------------------------------- batchnorm
def bn_stats(x):
    return _C.bn_stats(x)


def _count_args(count):
    if torch.is_tensor(count):
        return 0.0, count.reshape(1)
    return float(count), None


def bn_sum_partials(stats):
    return _C.bn_sum_partials(stats.contiguous()) if stats.dim() == 3 else stats


def bn_finalize(stats, count, eps: float, running_mean=None, running_var=None, momentum: float = 0.0,
                num_batches_tracked=None):
    """Partial rows [P,2,C] -> mean / invstd, + in-place running stats and num_batches_tracked:
    ONE tiny launch, no host sync."""
    c, cd = _count_args(count)
    if stats.dim() == 2:
        stats = stats[None]
    return _C.bn_finalize(stats.contiguous(), c, cd, eps, momentum, running_mean, running_var, num_batches_tracked,
                          None)[:2]


def bn_finalize_sync(stats, count: float, eps: float, running_mean, running_var, momentum: float, num_batches_tracked,
                     ipc):
    """SyncBatchNorm: like ``bn_finalize``, but the kernel exchanges ``(count, mean, M2)`` with the other ranks
    over the IPC mailboxes ``ipc`` (``_C.IpcAllReduce``) and merges them itself -> ``(mean, invstd, count_total)``
    with ``count_total`` a 1-element device tensor."""
    if stats.dim() == 2:
        stats = stats[None]
    return _C.bn_finalize(stats.contiguous(), float(count), None, eps, momentum, running_mean, running_var,
                          num_batches_tracked, ipc)


def bn_apply_act(x, mean, invstd, gamma, beta, residual=None, relu: bool = False, res_affine=None):
    """-> (z, act): ``act`` is the uint8 [M, C/8] ReLU bitmask the backward kernels read (None without relu).
    ``res_affine = (mean, invstd, gamma, beta)``: ``residual`` is the un-normalised input of another (ReLU-less)
    BatchNorm whose affine the kernel applies while it reads it."""
    ra = None if res_affine is None else [t.contiguous() for t in res_affine]
    return _C.bn_apply_act(x, mean, invstd, gamma.contiguous(), beta.contiguous(), residual, relu, ra)


def bn_bwd_reduce(dz, z, x, mean, invstd, relu: bool):
    return _C.bn_bwd_reduce(dz, z if relu else None, x, mean, invstd, relu)


def bn_bwd_elemt(dz, z, x, mean, invstd, gamma, psums, count, relu: bool, want_dres: bool, ipc=None):
    """``ipc`` (SyncBatchNorm): ``psums`` are this rank's partial rows, the kernel adds the other ranks' sums over the
    IPC mailboxes itself; the returned ``sums`` stay local."""
    c, cd = _count_args(count)
    if psums.dim() == 2:
        psums = psums[None]
    dx, dres, sums = _C.bn_bwd_elemt(dz, z if relu else None, x, mean, invstd, gamma.contiguous(),
                                     psums.contiguous(), c, cd, relu, want_dres, ipc)
    return dx, (dres if want_dres else None), sums


def bn_bwd_elemt2(dz, z, xa, mean_a, invstd_a, gamma_a, psums_a, xb, mean_b, invstd_b, gamma_b, psums_b, count, ipc=None):
    """Input gradients of two BatchNorms fed by the same gradient ``dz`` and ReLU mask ``z`` (bottleneck bn3 + the BN
    of the downsample branch) in one pass over ``dz`` -> ``(dxa, dxb, sums_a, sums_b)``."""
    c, cd = _count_args(count)
    pa = psums_a[None] if psums_a.dim() == 2 else psums_a
    pb = psums_b[None] if psums_b.dim() == 2 else psums_b
    return _C.bn_bwd_elemt2(dz, z, xa, mean_a, invstd_a, gamma_a.contiguous(), pa.contiguous(), xb, mean_b, invstd_b,
                            gamma_b.contiguous(), pb.contiguous(), c, cd, ipc)


# ---------------------------------------------------------------- attention
def attention_unsupported_reason(qkv, heads: int):
    if qkv.dtype != _BF16:
        return f"dtype {qkv.dtype} (native attention is bf16)"
    if qkv.dim() != 3 or qkv.shape[2] % (3 * heads) or qkv.shape[2] // (3 * heads) not in (32, 64, 128):
        return f"head_dim {qkv.shape[-1] / (3 * heads):g} (native attention: head_dim 32, 64 or 128)"
    return None


def attention_fwd(qkv, heads: int):
    """packed qkv [B, S, 3*H*hd] -> (out [B, S, H*hd], lse [B, H, S]); softmax scale 1/sqrt(hd), hd in {32, 64, 128};
    whole-head kernels for hd 64 up to 256 tokens, block-streaming kernels for everything else (any length)"""
    return _C.attention_fwd(qkv.contiguous(), heads)


def attention_bwd(dout, qkv, out, lse, heads: int):
    return _C.attention_bwd(dout.contiguous(), qkv, out, lse, heads)


# ---------------------------------------------------------------- layernorm
def layernorm_unsupported_reason(x):
    if x.dtype != _BF16:
        return f"dtype {x.dtype} (native LayerNorm is bf16)"
    if x.shape[-1] % 4 or x.shape[-1] > 2048:
        return f"width {x.shape[-1]} (native LayerNorm: multiples of 4 up to 2048)"
    return None


def layernorm_fwd(x, gamma, beta, eps: float):
    """-> (y, mean, rstd) over the last dimension"""
    return _C.layernorm_fwd(x.contiguous(), gamma.contiguous(), beta.contiguous(), eps)


def layernorm_bwd(dy, x, mean, rstd, gamma, addend=None):
    """-> (dx, dgamma, dbeta); ``addend`` (shape of x) is added to dx in the same pass"""
    return _C.layernorm_bwd(dy.contiguous(), x, mean, rstd, gamma.contiguous(),
                            None if addend is None else addend.contiguous())


# ------------------------------------------------------------------- linear
def linear_unsupported_reason(x, w, padded_ok: bool = False):
    """``x [..., K]`` against ``w [N, K]``: why the MFMA GEMM (csrc/kernels/gemm.hip) cannot take it, or None.
    ``padded_ok``: the caller zero-pads N to a multiple of 8 itself"""
    if x.dtype != _BF16 or w.dtype != _BF16:
        return f"dtype {x.dtype}/{w.dtype} (native GEMM is bf16)"
    n, k = w.shape
    if k % 8 or (n % 8 and not padded_ok):
        return f"features {k}->{n} not multiples of 8"
    return None


def _mat(t):
    """2-D view with contiguous 16-B-aligned rows (copies only when it has to)"""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1 or t2.stride(0) % 8 or t2.data_ptr() % 16:
        t2 = t2.contiguous()
    return t2


def gemm_nt(a, w, bias=None, residual=None, gelu_in=None, act: int = 0, want_preact: bool = False,
            want_colsum: bool = False):
    """``epilogue(a @ w^T)`` on the matrix cores -> ``(c, preact, colsum)``; a ``[M, K]``, w ``[N, K]`` bf16.
    bias fp32 (added before the rounding), ``act=1`` GELU, ``residual`` added last, ``gelu_in``: the result is
    multiplied by GELU'(gelu_in); ``colsum``: fp32 column sums of ``c``."""
    b32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float()).contiguous()
    k = a.shape[-1]
    if k % 64 and (residual is not None or gelu_in is not None or act):
        # the ragged-K build of the kernel carries no residual / GELU epilogue (hidden sizes of real transformers
        # are multiples of 64; this is a toy width): zero columns up to the next K-tile change nothing in the product
        pad = (0, 64 - k % 64)
        a, w = torch.nn.functional.pad(_mat(a), pad), torch.nn.functional.pad(_mat(w), pad)
    return _C.gemm_nt(_mat(a), _mat(w), b32, None if residual is None else _mat(residual),
                      None if gelu_in is None else _mat(gelu_in), act, want_preact, want_colsum)


def linear_wgrad(dy, x):
    """fp32 ``dy^T @ x`` -> ``[N, K]``"""
    return _C.linear_wgrad(dy.reshape(-1, dy.shape[-1]).contiguous(), x.reshape(-1, x.shape[-1]).contiguous())


def weight_t(w):
    """``w^T`` as a contiguous ``[K, N]`` bf16 matrix (the K-contiguous operand of the data-gradient GEMM); cached per
    weights epoch for parameters, refreshed together with the conv weights in one multi-tensor launch"""
    n, k = w.shape
    w4 = getattr(w, "_pdt_w4", None)
    if w4 is None:
        w4 = w.view(n, k, 1, 1)
        if getattr(w, "_pdt_param", False):
            w4._pdt_param = True
            w._pdt_w4 = w4
    return _transposed(w4).view(k, n)


def as_matrix_param(w, n: int):
    """``[n, kh*kw*cin]`` matrix in (kh, kw, cin) order of a conv weight (the patch embedding's weight as a GEMM
    operand): a view for channels_last storage; the view of a parameter's working copy keeps its identity across
    steps so that its transpose stays cached"""
    v = getattr(w, "_pdt_2d", None)
    if v is None:
        v = w.permute(0, 2, 3, 1).reshape(n, -1)
        if getattr(w, "_pdt_param", False) and v.data_ptr() == w.data_ptr():
            v._pdt_param = True
            w._pdt_2d = v
    return v


# ----------------------------------------------------------------- vit glue
def patchify(x, p: int):
    """channels_last bf16 image ``[B,3,S,S]`` -> ``[B*(S/p)^2, p*p*3]`` patches in (py, px, c) order"""
    return _C.patchify(x, p)


def patchify_unsupported_reason(x, p: int):
    if x.dtype != _BF16 or not _is_cl(x) or x.shape[1] != 3 or p % 8 or x.shape[2] != x.shape[3] or x.shape[2] % p:
        return f"image {tuple(x.shape)} {x.dtype} patch {p}"
    return None


def vit_embed_fwd(patch, cls, pos):
    """``cat(cls, patch) + pos`` -> ``[B, T+1, D]``"""
    return _C.vit_embed_fwd(patch.contiguous(), cls.reshape(-1).float().contiguous(), pos.reshape(-1).float().contiguous())


def vit_embed_bwd(dx):
    """-> (dpatch [B,T,D], dpos [T+1,D] fp32); the class-token gradient is ``dpos[0]``"""
    return _C.vit_embed_bwd(dx.contiguous())


# -------------------------------------------------------------------- pools
def maxpool3x3s2_fwd(x):
    """-> (y, argmax taps as uint8 [B,Ho,Wo,C])"""
    return _C.maxpool3x3s2_fwd(x)


def maxpool3x3s2_bwd(dy, x, y, argmax):
    return _C.maxpool3x3s2_bwd(dy, argmax, list(x.shape))


def bn_relu_maxpool_bwd_ok(u) -> bool:
    """The gather-based backward of the fused stem tail takes even H, W and < 2^32 elements."""
    return u.shape[2] % 2 == 0 and u.shape[3] % 2 == 0 and u.numel() < (1 << 32)


def bn_relu_maxpool_bwd(dy, argmax, u, mask, mean, invstd, gamma, count, ipc=None):
    """Backward of ``bn_relu_maxpool_fwd`` -> ``(du, sums [2,C])``: both BN-backward passes gather the pooled gradient
    themselves, the full-resolution gradient is never written."""
    c, cd = _count_args(count)
    return _C.bn_relu_maxpool_bwd(dy, argmax, u, mask, mean, invstd, gamma.contiguous(), c, cd, ipc)


def bn_relu_maxpool_fwd(u, mean, invstd, gamma, beta):
    """``maxpool3x3/2/1(relu(bn(u)))`` in one pass over the conv output -> (y, argmax taps uint8 [B,Ho,Wo,C],
    1-bit ReLU mask [B*H*W, C/8] of the full-resolution activation, which is never written)"""
    return _C.bn_relu_maxpool_fwd(u, mean, invstd, gamma.contiguous(), beta.contiguous())


def global_avgpool_fwd(x):
    return _C.global_avgpool_fwd(x)


def global_avgpool_bwd(dy, x_shape):
    return _C.global_avgpool_bwd(dy, list(x_shape))


# --------------------------------------------------------------------- loss
def colsum(x):
    """fp32 column sums of a 2-D tensor (Linear bias gradient): the HIP streaming kernel for contiguous bf16 rows
    of a multiple of 8 columns, ``sum(0)`` with fp32 accumulation otherwise."""
    if (x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous() and x.shape[1] % 8 == 0
            and x.data_ptr() % 16 == 0):
        return _C.colsum(x)
    return x.sum(0, dtype=torch.float32)


def softmax_ce_fwd_bwd(logits, labels):
    if logits.dtype not in (_BF16, torch.float32):
        return reference.softmax_ce_fwd_bwd(logits, labels)
    return _C.softmax_ce(logits, labels)


# ----------------------------------------------------------- flatten / cast
def _aligned(t):
    dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    return dense and t.data_ptr() % 16 == 0


def flatten_scaled_views(grads, views, scale: float):
    fast_g, fast_v = [], []
    for g, v in zip(grads, views):
        if (g.is_cuda and g.dtype in (torch.float32, _BF16) and v.dtype in (torch.float32, _BF16)
                and g.stride() == v.stride() and _aligned(g) and _aligned(v)):
            if g.data_ptr() == v.data_ptr() and scale == 1.0:
                continue
            fast_g.append(g)
            fast_v.append(v)
        else:
            reference.flatten_scaled_views([g], [v], scale)
    if fast_g:
        _C.multi_flatten(fast_g, fast_v, scale)


def nchw_to_nhwc_lowp(x, dtype=_BF16, scale: float = 1.0):
    if dtype != _BF16 or not x.is_contiguous() or x.dtype not in (torch.float32, torch.uint8) or (
            x.dtype == torch.float32 and x.shape[1] * (x.shape[3] + 1) * 4 > 64 * 1024):
        return reference.nchw_to_nhwc_lowp(x, dtype, scale)
    return _C.to_nhwc_bf16(x, scale)


# --------------------------------------------------------------- optimizers
_REFUSALS = ("wrong device/dtype", "must be dense", "16-B aligned", "layout != param layout", "multi-tensor ops take")


def _layout_refusal(e: Exception) -> bool:
    msg = str(e)
    return any(r in msg for r in _REFUSALS)


def multi_adam(params, grads, ms, vs, lps, lr, beta1, beta2, eps, weight_decay, step, adamw, grad_scale=1.0,
               step_dev=None):
    """``step_dev`` (fp32 device scalar holding the step count): the bias corrections are computed in the kernel
    from it, so a captured graph replays correctly; ``step`` is then only used by the fallback path."""
    # the extension validates every tensor (device, dtype, dense, 16-B aligned, layouts equal) before it launches
    # anything; a refusal sends the whole group through the per-tensor oracle path
    try:
        _C.multi_adam(list(params), list(grads), list(ms), list(vs), list(lps), lr, beta1, beta2, eps, weight_decay,
                      int(step), bool(adamw), grad_scale, step_dev)
    except RuntimeError as e:
        if not _layout_refusal(e):
            raise
        for p, g, m, v, lp in zip(params, grads, ms, vs, lps):
            reference.adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, float(step), adamw, grad_scale, lp)
    bump_weights_epoch()


def multi_sgd(params, grads, bufs, lps, lr, momentum, dampening, weight_decay, nesterov, first, grad_scale=1.0):
    try:
        _C.multi_sgd(list(params), list(grads), list(bufs), list(lps), lr, momentum, dampening, weight_decay,
                     bool(nesterov), bool(first), grad_scale)
    except RuntimeError as e:
        if not _layout_refusal(e):
            raise
        for p, g, b, lp in zip(params, grads, bufs, lps):
            reference.sgd_step(p, g, b, lr, momentum, dampening, weight_decay, nesterov, first, grad_scale, lp)
    bump_weights_epoch()
