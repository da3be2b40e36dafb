"""Pure-PyTorch oracles for every native op (same signatures as ``ops.native``).

These run anywhere (CPU, any layout, any dtype) and are (a) the numerics
oracle the GPU tests compare the HIP kernels against and (b) the compute path
of the CPU/gloo plumbing configuration.  They are never the shipped GPU path.

Conventions shared with the native ops: image tensors are logical ``[B,C,H,W]``
(native: channels_last memory = NHWC, bf16); per-channel vectors are fp32.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------- conv ----
def conv2d_fwd(x, w, stride: int, padding: int, want_stats: bool = False, return_saved: bool = False):
    y = F.conv2d(x, w.to(x.dtype), None, stride, padding)
    stats = None
    if want_stats:
        yf = y.float()
        stats = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))], 0)[None]  # partial rows [1, 2, C]
    return (y, stats, x) if return_saved else (y, stats)


def conv2d_dgrad(dy, w, x_shape, stride: int, padding: int, addend=None, addend_mask=None, addend_stride: int = 1):
    """``addend_mask``: activity "mask" of a BN+ReLU as this backend keeps it (the post-ReLU output itself);
    the addend only counts where that ReLU was active."""
    dx = _conv2d_dgrad(dy, w, x_shape, stride, padding)
    if addend is None:
        return dx
    if addend_stride == 2:  # compact gradient of a 1x1 stride-2 conv of the same input: lands on the even pixels
        dx = dx.clone()
        dx[:, :, ::2, ::2] += addend.to(dx.dtype)
        return dx
    if addend_mask is not None:
        addend = addend * (addend_mask > 0).to(addend.dtype)
    return dx + addend


def apply_relu_gate(g, mask):
    """``g`` where the ReLU that produced ``mask`` was active, 0 elsewhere."""
    return g * (mask > 0).to(g.dtype)


def _conv2d_dgrad(dy, w, x_shape, stride: int, padding: int):
    return torch.ops.aten.convolution_backward(
        dy, torch.empty(x_shape, dtype=dy.dtype, device=dy.device).expand(x_shape), w.to(dy.dtype), None,
        [stride, stride], [padding, padding], [1, 1], False, [0, 0], 1, [True, False, False])[0]


def conv2d_wgrad(dy, x, w_shape, stride: int, padding: int):
    gw = torch.ops.aten.convolution_backward(
        dy, x, torch.empty(w_shape, dtype=x.dtype, device=x.device), None,
        [stride, stride], [padding, padding], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    return gw.float()


# ------------------------------------------------------------ batchnorm ----
def bn_stats(x):
    """Per-channel ``[sum, sum_sq]`` in fp32 as partial rows ``[P, 2, C]`` (the oracle has P = 1;
    the HIP kernels emit one row per block / conv tile and the consumer sums them)."""
    xf = x.float()
    return torch.stack([xf.sum((0, 2, 3)), (xf * xf).sum((0, 2, 3))], 0)[None]


def bn_sum_partials(stats):
    """``[P, 2, C] -> [2, C]``"""
    return stats.sum(0) if stats.dim() == 3 else stats


def bn_finalize(stats, count, eps: float, running_mean=None, running_var=None, momentum: float = 0.0,
                num_batches_tracked=None):
    """partial sums ``[P,2,C]`` + element count (float or 1-element tensor) -> (mean, invstd) with the
    biased variance; updates the running stats (unbiased variance, torch semantics) and
    ``num_batches_tracked`` in place."""
    stats = bn_sum_partials(stats)
    if num_batches_tracked is not None:
        num_batches_tracked.add_(1)
    if torch.is_tensor(count):
        count = count.reshape(()).to(stats.dtype)
    mean = stats[0] / count
    var = (stats[1] / count - mean * mean).clamp_min(0.0)
    invstd = torch.rsqrt(var + eps)
    if running_mean is not None:
        denom = torch.clamp(count - 1.0, min=1.0) if torch.is_tensor(count) else max(count - 1.0, 1.0)
        unbiased = var * (count / denom)
        running_mean.mul_(1.0 - momentum).add_(mean.to(running_mean.dtype) * momentum)
        running_var.mul_(1.0 - momentum).add_(unbiased.to(running_var.dtype) * momentum)
    return mean, invstd


def bn_apply_act(x, mean, invstd, gamma, beta, residual=None, relu: bool = False, res_affine=None):
    """``z = act((x-mean)*invstd*gamma + beta [+ residual])`` in the input dtype -> ``(z, act)`` where
    ``act`` is what the backward ops take to gate the ReLU (here ``z`` itself; the HIP path hands out a
    1-bit-per-element mask instead)."""
    a = (invstd * gamma).view(1, -1, 1, 1)
    b = (beta - mean * invstd * gamma).view(1, -1, 1, 1)
    z = x.float() * a + b
    if residual is not None:
        r = residual.float()
        if res_affine is not None:  # the residual is the input of a ReLU-less BN: apply that BN's affine here
            rm, ri, rg, rb = res_affine
            r = r * (ri * rg).view(1, -1, 1, 1) + (rb - rm * ri * rg).view(1, -1, 1, 1)
        z = z + r
    if relu:
        z = z.clamp_min_(0.0)
    z = z.to(x.dtype)
    return z, (z if relu else None)


def bn_bwd_reduce(dz, z, x, mean, invstd, relu: bool):
    """-> partial rows ``[P, 2, C]`` of ``sum(dzm)``, ``sum(dzm * xhat)`` with ``dzm = dz * (z > 0)``."""
    g = dz.float()
    if relu:
        g = g * (z > 0)
    xhat = (x.float() - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    return torch.stack([g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))], 0)[None]


def bn_bwd_elemt(dz, z, x, mean, invstd, gamma, psums, count, relu: bool, want_dres: bool):
    """-> ``(dx, dres, sums)``; ``psums`` are partial rows ``[P,2,C]`` of ``bn_bwd_reduce`` (or the
    globally reduced sums as ``[1,2,C]``), ``count`` the global element count per channel and
    ``sums`` the summed ``[2,C]``."""
    sums = bn_sum_partials(psums)
    if torch.is_tensor(count):
        count = count.reshape(())
    g = dz.float()
    if relu:
        g = g * (z > 0)
    v = lambda t: t.view(1, -1, 1, 1)  # noqa: E731
    xhat = (x.float() - v(mean)) * v(invstd)
    dx = v(gamma * invstd) * (g - v(sums[0] / count) - xhat * v(sums[1] / count))
    dres = g.to(dz.dtype) if want_dres else None
    return dx.to(x.dtype), dres, sums


def bn_bwd_elemt2(dz, z, xa, mean_a, invstd_a, gamma_a, psums_a, xb, mean_b, invstd_b, gamma_b, psums_b, count, ipc=None):
    """Two BatchNorms fed by one gradient (``native.bn_bwd_elemt2``) -> ``(dxa, dxb, sums_a, sums_b)``."""
    relu = z is not None
    dxa, _, sa = bn_bwd_elemt(dz, z, xa, mean_a, invstd_a, gamma_a, psums_a, count, relu, False)
    dxb, _, sb = bn_bwd_elemt(dz, z, xb, mean_b, invstd_b, gamma_b, psums_b, count, relu, False)
    return dxa, dxb, sa, sb


# ------------------------------------------------------------ attention ----
def _split_heads(qkv, heads):
    b, s, w = qkv.shape
    hd = w // (3 * heads)
    t = qkv.view(b, s, 3, heads, hd).permute(2, 0, 3, 1, 4)  # [3, b, h, s, hd]
    return t[0], t[1], t[2], hd


def attention_fwd(qkv, heads: int):
    q, k, v, hd = _split_heads(qkv, heads)
    scores = torch.matmul(q.float() * (hd ** -0.5), k.float().transpose(-1, -2))
    lse = torch.logsumexp(scores, dim=-1)
    p = torch.exp(scores - lse[..., None])
    out = torch.matmul(p, v.float()).transpose(1, 2).reshape(qkv.shape[0], qkv.shape[1], heads * hd)
    return out.to(qkv.dtype), lse


def attention_bwd(dout, qkv, out, lse, heads: int):
    q, k, v, hd = _split_heads(qkv, heads)
    b, s, _ = qkv.shape
    scale = hd ** -0.5
    qf, kf, vf = q.float(), k.float(), v.float()
    do = dout.float().view(b, s, heads, hd).transpose(1, 2)
    o = out.float().view(b, s, heads, hd).transpose(1, 2)
    p = torch.exp(torch.matmul(qf * scale, kf.transpose(-1, -2)) - lse[..., None])
    dv = torch.matmul(p.transpose(-1, -2), do)
    dp = torch.matmul(do, vf.transpose(-1, -2))
    ds = p * (dp - (do * o).sum(-1, keepdim=True))
    dq = torch.matmul(ds, kf) * scale
    dk = torch.matmul(ds.transpose(-1, -2), qf) * scale
    dqkv = torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(b, s, 3 * heads * hd)
    return dqkv.to(qkv.dtype)


# ------------------------------------------------------------ layernorm ----
def layernorm_fwd(x, gamma, beta, eps: float):
    xf = x.float()
    mean = xf.mean(-1)
    var = xf.var(-1, unbiased=False)
    rstd = torch.rsqrt(var + eps)
    y = (xf - mean[..., None]) * rstd[..., None] * gamma + beta
    return y.to(x.dtype), mean.reshape(-1), rstd.reshape(-1)


def layernorm_bwd(dy, x, mean, rstd, gamma, addend=None):
    d = x.shape[-1]
    xf, dyf = x.float().reshape(-1, d), dy.float().reshape(-1, d)
    xhat = (xf - mean[:, None]) * rstd[:, None]
    g = dyf * gamma
    c1 = g.mean(-1, keepdim=True)
    c2 = (g * xhat).mean(-1, keepdim=True)
    dx = rstd[:, None] * (g - c1 - xhat * c2)
    if addend is not None:
        dx = dx + addend.float().reshape(-1, d)
    return dx.to(x.dtype).view_as(x), (dyf * xhat).sum(0), dyf.sum(0)


# ---------------------------------------------------------------- pools ----
def maxpool3x3s2_fwd(x):
    """-> (y, aux); the oracle needs no aux"""
    return F.max_pool2d(x, 3, 2, 1), None


def maxpool3x3s2_bwd(dy, x, y, aux=None):
    """Gradient routed to the first max in window scan order (torch semantics)."""
    with torch.enable_grad():
        xr = x.detach().float().requires_grad_(True)
        out = F.max_pool2d(xr, 3, 2, 1)
        (g,) = torch.autograd.grad(out, xr, dy.float())
    return g.to(x.dtype)


def bn_relu_maxpool_fwd(u, mean, invstd, gamma, beta):
    """Oracle of the fused stem tail: the same composition of ops the unfused path runs -> (y, aux, act) with
    ``act`` what the BN backward ops gate the ReLU with (here the full-resolution activation itself)."""
    z, act = bn_apply_act(u, mean, invstd, gamma, beta, None, True)
    return F.max_pool2d(z, 3, 2, 1), None, act


def global_avgpool_fwd(x):
    return x.float().mean((2, 3)).to(x.dtype)


def global_avgpool_bwd(dy, x_shape):
    b, c, h, w = x_shape
    return (dy.float() / (h * w)).to(dy.dtype).view(b, c, 1, 1).expand(b, c, h, w).contiguous(
        memory_format=torch.channels_last)


# ----------------------------------------------------------------- loss ----
def colsum(x):
    return x.sum(0, dtype=torch.float32)


def softmax_ce_fwd_bwd(logits, labels):
    """Mean CE loss (fp32 scalar) and ``dlogits = (softmax - onehot)/B`` in the
    logits dtype, in one pass."""
    lf = logits.float()
    lse = torch.logsumexp(lf, 1)
    picked = lf.gather(1, labels.view(-1, 1)).squeeze(1)
    loss = (lse - picked).mean()
    p = torch.softmax(lf, 1)
    p.scatter_add_(1, labels.view(-1, 1), torch.full_like(picked, -1.0).view(-1, 1))
    return loss, (p / logits.shape[0]).to(logits.dtype)


# ---------------------------------------------------------------- linear ---
def linear_unsupported_reason(x, w, padded_ok: bool = False):
    return None


def gemm_nt(a, w, bias=None, residual=None, gelu_in=None, act: int = 0, want_preact: bool = False,
            want_colsum: bool = False):
    """Oracle of csrc/kernels/gemm.hip: fp32 math, every intermediate rounded where the kernel rounds it"""
    a2, w2 = a.reshape(-1, a.shape[-1]), w.reshape(-1, w.shape[-1])
    dt = a.dtype
    c = a2.float() @ w2.float().t()
    if bias is not None:
        c = c + bias.float()
    c = c.to(dt)
    pre = c if want_preact else None
    if act == 1:
        c = F.gelu(c.float()).to(dt)
    if gelu_in is not None:
        u = gelu_in.reshape(c.shape).float()
        cdf = 0.5 * (1.0 + torch.erf(u * 0.7071067811865476))
        pdf = torch.exp(-0.5 * u * u) * 0.3989422804014327
        c = (c.float() * (cdf + u * pdf)).to(dt)
    if residual is not None:
        c = (c.float() + residual.reshape(c.shape).float()).to(dt)
    return c, pre, (c.float().sum(0) if want_colsum else None)


def linear_wgrad(dy, x):
    return dy.reshape(-1, dy.shape[-1]).float().t() @ x.reshape(-1, x.shape[-1]).float()


def weight_t(w):
    return w.t().contiguous()


def as_matrix_param(w, n: int):
    return w.permute(0, 2, 3, 1).reshape(n, -1)


def patchify(x, p: int):
    b, c, h, w = x.shape
    g = h // p
    return x.permute(0, 2, 3, 1).reshape(b, g, p, g, p, c).permute(0, 1, 3, 2, 4, 5).reshape(b * g * g, p * p * c)


def patchify_unsupported_reason(x, p: int):
    return None


def vit_embed_fwd(patch, cls, pos):
    b, t, d = patch.shape
    x = torch.cat([cls.reshape(1, 1, d).float().expand(b, 1, d), patch.float()], 1) + pos.reshape(1, t + 1, d).float()
    return x.to(patch.dtype)


def vit_embed_bwd(dx):
    return dx[:, 1:].contiguous(), dx.float().sum(0)


def linear_fwd(x, w, bias):
    return F.linear(x, w.to(x.dtype), None if bias is None else bias.to(x.dtype))


def linear_bwd(dy, x, w):
    dx = dy @ w.to(dy.dtype)
    dw = (dy.float().t() @ x.float())
    db = dy.float().sum(0)
    return dx, dw, db


# ------------------------------------------------------------ optimizers ---
def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, adamw: bool, grad_scale: float = 1.0,
              p_lp=None):
    g = g.float() * grad_scale
    if weight_decay != 0.0:
        if adamw:
            p.mul_(1.0 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    if p_lp is not None:
        p_lp.copy_(p)


def sgd_step(p, g, buf, lr, momentum, dampening, weight_decay, nesterov: bool, first: bool, grad_scale: float = 1.0,
             p_lp=None):
    g = g.float() * grad_scale
    if weight_decay != 0.0:
        g = g + weight_decay * p
    if momentum != 0.0:
        if first:
            buf.copy_(g)
        else:
            buf.mul_(momentum).add_(g, alpha=1.0 - dampening)
        g = g + momentum * buf if nesterov else buf
    p.add_(g, alpha=-lr)
    if p_lp is not None:
        p_lp.copy_(p)


# ------------------------------------------------------- flatten / cast ----
def flatten_scaled(tensors, flat, offsets, scale: float):
    """``flat[off_i : off_i+n_i] = tensors[i].flatten() * scale`` (dtype of ``flat``)."""
    for t, off in zip(tensors, offsets):
        flat[off: off + t.numel()].copy_((t.reshape(-1).float() * scale))


def unflatten(flat, tensors, offsets):
    for t, off in zip(tensors, offsets):
        t.copy_(flat[off: off + t.numel()].view_as(t))


def nchw_to_nhwc_lowp(x, dtype=torch.bfloat16, scale: float = 1.0):
    """fp32/uint8 NCHW (or NHWC uint8 when ``x.dim()==4 and x.dtype==uint8 and last dim<=4``)
    -> channels_last ``dtype`` with an optional scale (1/255 for uint8)."""
    if x.dtype == torch.uint8 and x.shape[-1] <= 4:  # NHWC uint8 block from the loader
        x = x.permute(0, 3, 1, 2)
    return (x.float() * scale).to(dtype).contiguous(memory_format=torch.channels_last)


def flatten_scaled_views(grads, views, scale: float):
    """``views[i] <- grads[i] * scale`` (cast to the view dtype).  ``views[i]`` has the
    gradient's logical shape; aliasing (``views[i] is grads[i]``'s memory) is allowed."""
    for g, v in zip(grads, views):
        if g.data_ptr() == v.data_ptr() and g.stride() == v.stride():
            if scale != 1.0:
                v.mul_(scale)
        else:
            v.copy_(g.float() * scale if scale != 1.0 else g)
