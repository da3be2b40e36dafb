"""Autograd functions over the op layer (HIP kernels on GPU, oracles on CPU).

One training step of the reference (``/root/reference/main.py:100-104``) runs
53 conv + 53 BN + 49 ReLU + 16 residual adds + pools + fc + CE through stock
ATen/cuDNN kernels.  Here the graph is coarser: every
``conv -> BN -> (+residual) -> ReLU`` group is two autograd nodes

* ``conv2d``      implicit-GEMM conv; its epilogue also emits the per-channel
                  ``[sum, sum_sq]`` BN statistics of its own output (no extra
                  pass over the activation);
* ``bn_act``      normalise + affine (+ residual add) (+ ReLU) in one pass;
                  backward fuses the ReLU mask into the BN reductions.

SyncBatchNorm is the same node with one all-reduce of ``[sum, sum_sq, count]``
in forward and one of ``[sum_dy, sum_dy_xhat]`` in backward (all-reduce-of-sums
formulation: no host sync, so the step stays graph-capturable; SURVEY §2.5
S4a/S4b).  Works on CPU tensors over gloo, which stock SyncBatchNorm refuses.
"""
from __future__ import annotations

import collections
import os

import torch
import torch.distributed as tdist
import torch.nn as tnn

from ..ops import backend_for, reference

fallback_counts: collections.Counter = collections.Counter()
# PDT_PLAIN_TORCH=1: run conv/BN/ReLU through stock PyTorch-ROCm ops (same-hardware calibration baseline)
_PLAIN_TORCH = os.environ.get("PDT_PLAIN_TORCH", "0") == "1"


def _note_fallback(op: str, why: str) -> None:
    fallback_counts[f"{op}: {why}"] += 1


def _cl(t: torch.Tensor) -> torch.Tensor:
    """channels_last-contiguous view/copy of a 4-D GPU tensor (no-op on CPU)."""
    if t.is_cuda and t.dim() == 4 and not t.is_contiguous(memory_format=torch.channels_last):
        return t.contiguous(memory_format=torch.channels_last)
    return t


# --------------------------------------------------------------------- conv
# ---- lazily gated residual gradients --------------------------------------------------------------------------
# The backward of ``relu(bn(y) + identity)`` owes ``identity`` the gradient ``dz * [relu active]``.  Writing that
# tensor costs a quarter of the kernel's traffic, and its only reader is the node that produced ``identity``: a conv
# node (which adds it inside its data-gradient epilogue) or the downsample BN (whose backward kernels know how to
# gate with a ReLU mask anyway).  So, when the model opts in, the BN node hands back the UNGATED ``dz`` tagged with
# the mask, and the producer applies the gate while it reads.  The producer is told at forward time to expect the
# tag (``ctx.expect_gate``) and raises if a gradient arrives without it: a lost tag can never go unnoticed.
_GATE_ATTR = "_pdt_relu_gate"
_GATE_AWARE = ("_Conv2dFnBackward", "_BNActFnBackward")


def _tag_gate(dz, mask):
    setattr(dz, _GATE_ATTR, (mask, dz.data_ptr(), dz._version))
    return dz


def _take_gate(g, expected: bool):
    """The mask ``g`` must be gated with (None: plain gradient).  Raises when a promised tag is missing/stale."""
    tag = getattr(g, _GATE_ATTR, None) if g is not None else None
    if tag is None:
        if expected and g is not None:
            raise RuntimeError("lazy residual gradient: the gradient arrived without its ReLU gate "
                               "(the residual tensor must have exactly one consumer; pass lazy_residual_grad=False)")
        return None
    mask, ptr, ver = tag
    if g.data_ptr() != ptr or g._version != ver:
        raise RuntimeError("lazy residual gradient: tagged gradient was modified before its consumer ran")
    return mask


class _Conv2dFn(torch.autograd.Function):
    """``(y, stats, skip)``: ``skip`` (when requested) is the input passed through, to be used as the
    residual branch of the block.  Its gradient then arrives HERE and is added inside the data-gradient
    kernel's epilogue, instead of autograd summing two activation-sized tensors with a separate add."""

    @staticmethod
    def forward(ctx, x, w, w_lp, stride, padding, want_stats, with_skip, compact_to=None):
        wk = w_lp if w_lp is not None else w
        be = backend_for(x)
        if be is not reference:
            why = be.conv_unsupported_reason(x, wk, stride, padding)
            if why is not None:
                _note_fallback("conv2d", why)
                be = reference
                want_stats = False  # the BN node's native one-pass bn_stats beats torch reductions over y
        # the BN+ReLU that produced x (if any) lets this node's data-gradient kernel do its backward reduction
        sink = getattr(x, "_pdt_bn_sink", None)
        x = _cl(x)
        if sink is not None and not (be is not reference and stride == 1 and sink.usable_for(x)):
            sink = None
        ctx.bn_sink = sink
        # 1x1 stride-2 conv on the skip of another conv node (downsample branch): its data gradient only touches
        # the even pixels, so it is handed to that node COMPACT (side channel, see backward) instead of as a
        # zero-filled full-size tensor that the other node would read back
        ctx.compact_to = compact_to
        y, stats, saved = be.conv2d_fwd(x, wk, stride, padding, want_stats, True)
        ctx.set_materialize_grads(False)  # no zero-filled gradient for the (non-differentiable) stats output
        ctx.save_for_backward(saved, wk)  # `saved` is x, or the packed image of the native 3-channel stem
        ctx.cfg = (be, stride, padding, tuple(w.shape), w.dtype, tuple(x.shape))
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats, (x.view_as(x) if with_skip else None)

    @staticmethod
    def backward(ctx, dy, _dstats, dskip):
        x, wk = ctx.saved_tensors
        be, stride, padding, w_shape, w_dtype, x_shape = ctx.cfg
        dx = dw = None
        compact = ctx.__dict__.pop("strided_addend", None)  # left here by the downsample conv's backward
        if compact is not None and dskip is not None:
            raise RuntimeError("compact skip gradient: the skip tensor also received a dense gradient")
        gate = _take_gate(dskip, getattr(ctx, "expect_gate", False))
        if dy is None:
            return (dskip if gate is None else be.apply_relu_gate(_cl(dskip), gate)), None, None, None, None, None, None, None
        dy = _cl(dy)
        if ctx.needs_input_grad[0] and ctx.compact_to is not None:
            b, _, ho, wo = dy.shape
            ctx.compact_to.strided_addend = be.conv2d_dgrad(dy, wk, (b, x_shape[1], ho, wo), 1, 0)
        elif ctx.needs_input_grad[0]:
            sink = ctx.bn_sink
            dskip = None if dskip is None else _cl(dskip)
            if compact is not None:
                if sink is not None:
                    dx, psums = be.conv2d_dgrad(dy, wk, x_shape, stride, padding, compact, sink.operands(), None, 2)
                    sink.fulfil(dx, psums)
                else:
                    dx = be.conv2d_dgrad(dy, wk, x_shape, stride, padding, compact, addend_stride=2)
            elif sink is not None and sink.second is not None and be is not reference:
                dx, psums, psums2 = be.conv2d_dgrad(dy, wk, x_shape, stride, padding, dskip, sink.operands(), gate,
                                                    bn_sink2=sink.second)
                sink.fulfil(dx, psums, psums2)
            elif sink is not None:
                dx, psums = be.conv2d_dgrad(dy, wk, x_shape, stride, padding, dskip, sink.operands(), gate)
                sink.fulfil(dx, psums)
            elif gate is not None:
                dx = be.conv2d_dgrad(dy, wk, x_shape, stride, padding, dskip, addend_mask=gate)
            else:
                dx = be.conv2d_dgrad(dy, wk, x_shape, stride, padding, dskip)
        if ctx.needs_input_grad[1]:
            dw = be.conv2d_wgrad(dy, x, w_shape, stride, padding)
            if dw.dtype != w_dtype:
                dw = dw.to(w_dtype)
        return dx, dw, None, None, None, None, None, None


def conv2d(x, weight, stride: int = 1, padding: int = 0, want_stats: bool = False, with_skip: bool = False,
           compact_skip_grad: bool = False):
    """Bias-free square-stride conv.  Returns ``(y, stats)`` (+ ``skip`` when ``with_skip``) where ``stats``
    is the fp32 partial rows ``[P, 2, Cout]`` of the per-channel ``[sum, sum_sq]`` of ``y`` (or ``None``)."""
    w_lp = getattr(weight, "_lp", None)
    if w_lp is not None and w_lp.dtype != x.dtype:
        w_lp = None
    compact_to = None
    if (compact_skip_grad and int(stride) == 2 and int(padding) == 0 and tuple(weight.shape[2:]) == (1, 1)
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and torch.is_grad_enabled() and x.requires_grad
            and getattr(x, "_pdt_lazy_grad_ok", False) and type(x.grad_fn).__name__ == "_Conv2dFnBackward"):
        compact_to = x.grad_fn  # caller's promise: x (a stride-1 conv node's skip output) has no other consumer
    y, stats, skip = _Conv2dFn.apply(x, weight, w_lp, int(stride), int(padding), bool(want_stats), bool(with_skip),
                                     compact_to)
    if with_skip and skip is not None:
        skip._pdt_lazy_grad_ok = int(stride) == 1  # the masked-addend epilogue exists for stride-1 data gradients
    return (y, stats, skip) if with_skip else (y, stats)


# ---------------------------------------------------------------- batchnorm
class BNConfig:
    """Per-layer BN state handed to the autograd node (not tensors of the graph)."""

    __slots__ = ("running_mean", "running_var", "num_batches_tracked", "momentum", "eps", "training",
                 "sync", "group")

    def __init__(self, running_mean, running_var, num_batches_tracked, momentum, eps, training, sync=False,
                 group=None):
        self.running_mean, self.running_var = running_mean, running_var
        self.num_batches_tracked = num_batches_tracked
        self.momentum, self.eps, self.training = momentum, eps, training
        self.sync, self.group = sync, group


def _sync_world(cfg: BNConfig) -> int:
    if not cfg.sync or not (tdist.is_available() and tdist.is_initialized()):
        return 1
    return tdist.get_world_size(cfg.group)


def _small_all_reduce(t: torch.Tensor, group) -> None:
    """SUM of a SyncBN statistics vector over the group, on the current stream: the one-kernel IPC path on GPUs
    that share a host, the process group's all-reduce elsewhere (``parallel/ipc.py``)."""
    if t.is_cuda:
        from ..parallel.ipc import small_all_reduce

        small_all_reduce(t, group)
    else:
        tdist.all_reduce(t, group=group)


class _BNSink:
    """Hand-over between a BN+ReLU node and the conv node that consumes its output.  The conv's data-gradient
    kernel produces exactly the tensor this BN's backward receives as ``dz``; given the BN's saved operands it can
    reduce ``[sum(dz*m), sum(dz*m*xhat)]`` in its epilogue while the tile is still in registers, and the BN backward
    then skips its own reduction pass (one full read of dz and x per layer).  The BN only trusts the result when
    the gradient it receives is that very tensor."""

    __slots__ = ("x", "mask", "mean", "invstd", "psums", "dz_ptr", "dz_version", "second", "psums2")

    def __init__(self):
        self.x = self.mask = self.mean = self.invstd = self.psums = None
        self.dz_ptr = self.dz_version = None
        self.second = self.psums2 = None

    def arm(self, x, mask, mean, invstd):
        self.x, self.mask, self.mean, self.invstd = x, mask, mean, invstd

    def arm_second(self, x2, mean2, invstd2):
        """A second BN that receives the same gradient through the same gate (downsample BN next to bn3)."""
        self.second = (x2, mean2, invstd2)

    def usable_for(self, z) -> bool:
        return self.x is not None and self.mask is not None and tuple(z.shape) == tuple(self.x.shape)

    def operands(self):
        return (self.x, self.mask, self.mean, self.invstd)

    def fulfil(self, dz, psums, psums2=None) -> None:
        self.psums, self.dz_ptr, self.dz_version = psums, dz.data_ptr(), dz._version
        self.psums2 = psums2

    def take(self, dz):
        """The fused partial sums if ``dz`` is the tensor they were computed from, else None (``take2``: the second
        BN's, call it first)."""
        ps = self.psums
        ok = ps is not None and dz.data_ptr() == self.dz_ptr and dz._version == self.dz_version
        self.x = self.mask = self.mean = self.invstd = self.psums = None  # drop the references either way
        self.second = self.psums2 = None
        return ps if ok else None

    def take2(self, dz):
        ps = self.psums2
        ok = ps is not None and dz.data_ptr() == self.dz_ptr and dz._version == self.dz_version
        return ps if ok else None


def _inkernel_exchange(be, x, cfg: BNConfig, n_floats: int):
    """The IPC mailboxes of ``cfg.group`` when the BN kernels can do the SyncBN exchange themselves (native
    kernels, ranks on one host, message fits), else None (``parallel/ipc.py`` decides once per group)."""
    if be is reference or not x.is_cuda or os.environ.get("PDT_SYNCBN_INKERNEL", "1") == "0":
        return None
    from ..parallel.ipc import small_all_reduce_path

    ipc, _ = small_all_reduce_path(cfg.group, x.device)
    return ipc if (ipc is not None and n_floats <= ipc.capacity) else None


def _bn_moments(be, x, stats, cfg: BNConfig):
    """Batch (or running) mean / invstd of ``x`` for one BN node -> ``(mean, invstd, count, count_t, world, ipc)``.
    In training with SyncBN the statistics are exchanged INSIDE the finalize kernel over the IPC mailboxes (``ipc``
    is then what the backward kernels use too); without them the packed ``[sum, sum_sq, count]`` is all-reduced
    once by the process group (no host sync either way)."""
    world = _sync_world(cfg)
    b, c, h, w = x.shape
    count = float(b * h * w)
    count_t = None
    ipc = None
    if cfg.training:
        if stats is None:
            stats = be.bn_stats(x)
        if world > 1:
            ipc = _inkernel_exchange(be, x, cfg, 3 * c + 1)
        mom = 0.0
        if cfg.running_mean is not None:
            mom = cfg.momentum
            if mom is None:  # cumulative moving average (host sync, like the stock module)
                mom = 1.0 / (float(cfg.num_batches_tracked.item()) + 1.0)
        if ipc is not None:
            # (count, mean, M2) of every rank merged in rank order by the kernel itself: zero extra launches
            mean, invstd, count_t = be.bn_finalize_sync(stats, count, cfg.eps, cfg.running_mean, cfg.running_var, mom,
                                                        cfg.num_batches_tracked if cfg.running_mean is not None else None,
                                                        ipc.native)
            return mean, invstd, count, count_t, world, ipc
        if world > 1:
            # [sum(C), sum_sq(C), count]: one all-reduce, counts may differ per rank
            packed = torch.empty(2 * c + 4, dtype=torch.float32, device=x.device)
            packed[: 2 * c].copy_(be.bn_sum_partials(stats).reshape(-1))
            packed[2 * c:] = count
            _small_all_reduce(packed, cfg.group)
            stats = packed[: 2 * c].view(1, 2, c)
            count_t = packed[2 * c: 2 * c + 1]
        cnt = count_t if count_t is not None else count
        mean, invstd = be.bn_finalize(stats, cnt, cfg.eps, cfg.running_mean, cfg.running_var, mom,
                                      cfg.num_batches_tracked if cfg.running_mean is not None else None)
    else:
        mean = cfg.running_mean.float()
        invstd = torch.rsqrt(cfg.running_var.float() + cfg.eps)
    return mean, invstd, count, count_t, world, None


class _PendingAffine:
    """A ReLU-less BN whose ONLY consumer is the residual input of another BN node (the downsample branch) does not
    have to write its output: it returns its input untouched together with this holder, and the consumer applies
    ``(mean, invstd, gamma, beta)`` while it reads the tensor (``bn_apply_act(..., res_affine=...)``).  The backward
    of the deferring node is unchanged: the gradient w.r.t. its (virtual) output is what the consumer returns for
    its residual.  ``consumed`` is checked in that backward: an un-normalised tensor that went anywhere else is an
    error, not a silently wrong activation.

    ``owner`` / ``result`` serve the backward: the consumer receives the gradient that is ALSO the deferring node's
    gradient (gated by the consumer's ReLU mask), so it computes both input gradients in one pass over it
    (``bn_bwd_elemt2``) and leaves the deferring node's share in ``result``."""

    __slots__ = ("operands", "consumed", "owner", "result")

    def __init__(self):
        self.operands, self.consumed = None, False
        self.owner = self.result = None


class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, stats, cfg: BNConfig, relu: bool, sink=None, lazy_res: bool = False,
                defer=None, res_affine=None, res_pend=None):
        be = backend_for(x)
        x = _cl(x)
        if residual is not None:
            residual = _cl(residual)
        if be is not reference:
            why = be.bn_unsupported_reason(x)
            if why is not None:
                _note_fallback("bn_act", why)
                be = reference
        mean, invstd, count, count_t, world, ctx.ipc = _bn_moments(be, x, stats, cfg)
        c = x.shape[1]
        g32 = gamma.float() if gamma is not None else torch.ones(c, device=x.device)
        b32 = beta.float() if beta is not None else torch.zeros(c, device=x.device)
        ctx.defer = defer
        if defer is not None:  # (relu False, no residual: checked by bn_act) the consumer applies the affine
            defer.operands = (mean, invstd, g32, b32)
            defer.owner = (x, mean, invstd, g32, count, world, ctx.ipc, be, cfg.training)
            z, act = x.view_as(x), None
        elif res_affine is not None:
            z, act = be.bn_apply_act(x, mean, invstd, g32, b32, residual, relu, res_affine)
        else:
            z, act = be.bn_apply_act(x, mean, invstd, g32, b32, residual, relu)
        ctx.res_pend = res_pend if res_affine is not None else None
        ctx.sink = None
        if sink is not None and relu and be is not reference and act is not None and act.dtype == torch.uint8:
            sink.arm(x, act, mean, invstd)
            ctx.sink = sink
            pend = ctx.res_pend
            if (_FUSE_DUAL_SUMS and DUAL_BN_BACKWARD and pend is not None and pend.owner is not None and cfg.training
                    and pend.owner[0].shape == x.shape and pend.owner[0].dtype == x.dtype and pend.owner[7] is be):
                sink.arm_second(pend.owner[0], pend.owner[1], pend.owner[2])  # the downsample BN: same dz, same gate
        ctx.lazy_res = bool(lazy_res and relu and residual is not None and act is not None)
        ctx.save_for_backward(x, act if relu else None, mean, invstd, g32, count_t)
        ctx.cfg = (be, cfg, relu, world, count, residual is not None,
                   None if gamma is None else gamma.dtype, None if beta is None else beta.dtype)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, z, mean, invstd, g32, count_t = ctx.saved_tensors
        be, cfg, relu, world, count, has_res, gdt, bdt = ctx.cfg
        if ctx.defer is not None and not ctx.defer.consumed:
            raise RuntimeError("deferred BatchNorm apply: the un-normalised output was not consumed as the residual "
                               "of a fused BN node (pass defer_apply=False)")
        psums2 = ctx.sink.take2(dz) if ctx.sink is not None else None  # (the downsample BN's, when the epilogue took both)
        psums = ctx.sink.take(dz) if ctx.sink is not None else None  # reduced by the consumer conv's dgrad epilogue
        gate = _take_gate(dz, getattr(ctx, "expect_gate", False))
        dz = _cl(dz)
        if ctx.defer is not None:
            done, ctx.defer.result, ctx.defer.owner = ctx.defer.result, None, None
            if done is not None and done[2] == dz.data_ptr() and done[3] == dz._version:
                # the consumer's backward already produced this node's input gradient (bn_bwd_elemt2)
                dx, sums = done[0], done[1]
                dgamma = sums[1].to(gdt) if (gdt is not None and ctx.needs_input_grad[1]) else None
                dbeta = sums[0].to(bdt) if (bdt is not None and ctx.needs_input_grad[2]) else None
                return (dx if ctx.needs_input_grad[0] else None), dgamma, dbeta, None, None, None, None, None, None, None, None, None
        if gate is not None:
            if relu:  # (not produced by the models here: a gated gradient into a BN that has its own ReLU)
                dz = be.apply_relu_gate(dz, gate)
            else:     # downsample BN: the block output's ReLU gates its gradient; the kernels take a mask anyway
                relu, z = True, gate
        if psums is None:
            psums = be.bn_bwd_reduce(dz, z, x, mean, invstd, relu)  # local partial rows [P, 2, C]
        want_res = has_res and ctx.needs_input_grad[3]
        lazy = want_res and ctx.lazy_res and gate is None
        want_dres = want_res and not lazy
        pend, other = ctx.res_pend, None
        if (DUAL_BN_BACKWARD and pend is not None and pend.owner is not None and lazy and relu and cfg.training
                and ctx.needs_input_grad[0] and z is not None and (world == 1 or ctx.ipc is not None)):
            ox, omean, oinvstd, og32, ocount, oworld, oipc, obe, otraining = pend.owner
            if (obe is be and otraining and oworld == world and ocount == count and oipc is ctx.ipc
                    and ox.shape == x.shape and ox.dtype == x.dtype):
                other = (ox, omean, oinvstd, og32)
        if other is not None:
            # relu(bn(x) + bn_ds(x_ds)): the downsample BN receives this same dz with this same gate; one pass over
            # dz and the mask produces both input gradients, its backward then only hands the result on
            ox, omean, oinvstd, og32 = other
            opsums = psums2 if (psums2 is not None and gate is None) else be.bn_bwd_reduce(dz, z, ox, omean, oinvstd, True)
            dx, odx, sums, osums = be.bn_bwd_elemt2(dz, z, x, mean, invstd, g32, psums, ox, omean, oinvstd, og32,
                                                     opsums, count_t if world > 1 else count,
                                                     ipc=ctx.ipc.native if world > 1 else None)
            pend.result = (odx, osums, dz.data_ptr(), dz._version)
            dres = None
        elif cfg.training and world > 1 and ctx.ipc is not None:
            # the kernel exchanges the two sums with the peers itself; `sums` stays local (DDP averages the affine grads)
            dx, dres, sums = be.bn_bwd_elemt(dz, z, x, mean, invstd, g32, psums, count_t, relu, want_dres,
                                             ipc=ctx.ipc.native)
        elif cfg.training and world > 1:
            # the affine grads stay local (DDP averages them); the sums that feed grad_input are global
            sums = be.bn_sum_partials(psums)
            gsums = sums.clone()
            _small_all_reduce(gsums, cfg.group)
            dx, dres, _ = be.bn_bwd_elemt(dz, z, x, mean, invstd, g32, gsums.view(1, 2, -1), count_t, relu, want_dres)
        elif cfg.training:
            dx, dres, sums = be.bn_bwd_elemt(dz, z, x, mean, invstd, g32, psums, count, relu, want_dres)
        else:
            # eval-mode BN: statistics are constants, no reduction terms in grad_input
            sums = be.bn_sum_partials(psums)
            dx, dres, _ = be.bn_bwd_elemt(dz, z, x, mean, invstd, g32, torch.zeros_like(sums).view(1, 2, -1), 1.0,
                                          relu, want_dres)
        dgamma = sums[1].to(gdt) if (gdt is not None and ctx.needs_input_grad[1]) else None
        dbeta = sums[0].to(bdt) if (bdt is not None and ctx.needs_input_grad[2]) else None
        if not ctx.needs_input_grad[0]:
            dx = None
        if lazy:
            dres = _tag_gate(dz, z)  # ungated gradient + this node's ReLU mask: the producer of the residual gates it
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None


DUAL_BN_BACKWARD = os.environ.get("PDT_DUAL_BN_BACKWARD", "1") != "0"  # bn3 + downsample BN in one pass over dz
_FUSE_BN_REDUCE = os.environ.get("PDT_FUSE_BN_REDUCE", "1") != "0"
_FUSE_DUAL_SUMS = os.environ.get("PDT_FUSE_DUAL_BN_SUMS", "1") != "0"  # ... and the downsample BN's sums in that epilogue too
_FUSE_STEM_POOL_BWD = os.environ.get("PDT_FUSE_STEM_POOL_BWD", "1") != "0"  # stem tail backward without dz


def bn_act(x, gamma, beta, cfg: BNConfig, residual=None, relu: bool = False, stats=None,
           lazy_residual_grad: bool = False, defer_apply: bool = False):
    """``lazy_residual_grad=True`` (model opt-in; requires that ``residual`` has NO other consumer): the gradient of
    ``residual`` is handed to its producer ungated together with this node's ReLU mask, see ``_tag_gate``."""
    if defer_apply and not relu and residual is None:
        # the caller promises that the result only feeds the ``residual`` of another bn_act (see _PendingAffine)
        pend = _PendingAffine()
        z = _BNActFn.apply(x, gamma, beta, None, stats, cfg, False, None, False, pend)
        z._pdt_pending_affine = pend
        if z.requires_grad:
            z._pdt_lazy_grad_ok = True
        return z
    res_affine = None
    pend = getattr(residual, "_pdt_pending_affine", None) if residual is not None else None
    if pend is not None:
        res_affine, pend.consumed = pend.operands, True
    lazy = False
    if lazy_residual_grad and relu and residual is not None and torch.is_grad_enabled() and residual.requires_grad:
        prod = residual.grad_fn
        if prod is not None and type(prod).__name__ in _GATE_AWARE and getattr(residual, "_pdt_lazy_grad_ok", False):
            lazy = True
    if relu and _FUSE_BN_REDUCE and x.is_cuda and torch.is_grad_enabled():
        sink = _BNSink()
        z = _BNActFn.apply(x, gamma, beta, residual, stats, cfg, True, sink, lazy, None, res_affine, pend)
        if sink.x is not None:
            z._pdt_bn_sink = sink  # picked up by the conv node that consumes z
    else:
        z = _BNActFn.apply(x, gamma, beta, residual, stats, cfg, bool(relu), None, lazy, None, res_affine, pend)
    if lazy and z.grad_fn is not None and getattr(z.grad_fn, "lazy_res", False):
        residual.grad_fn.expect_gate = True  # a gradient without its gate is now an error, not a wrong result
    if not relu and z.requires_grad:
        z._pdt_lazy_grad_ok = True  # a BN without ReLU can take a gated gradient (its kernels accept a mask)
    return z


# ---------------------------------------------------------------- attention
class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads):
        be = backend_for(qkv)
        if be is not reference:
            why = be.attention_unsupported_reason(qkv, heads)
            if why is not None:
                _note_fallback("attention", why)
                be = reference
        qkv = qkv.contiguous()
        out, lse = be.attention_fwd(qkv, heads)
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (be, heads)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        be, heads = ctx.cfg
        return be.attention_bwd(dout, qkv, out, lse, heads), None


def attention(qkv, heads: int):
    """Multi-head self-attention on the packed in-projection output ``[B, S, 3*D]`` (q | k | v, heads inside
    each) -> ``[B, S, D]``; scale 1/sqrt(head_dim), no mask, no dropout.  On bf16 GPU tensors with head_dim 64
    one fused HIP kernel per direction keeps the S x S scores on chip."""
    return _AttentionFn.apply(qkv, int(heads))


# ---------------------------------------------------------------- layernorm
class _LayerNormFn(torch.autograd.Function):
    """``(y, skip)``: ``skip`` (when requested) is the input passed through, to be used as the residual branch of a
    pre-LN transformer block.  Its gradient then arrives HERE and is added inside the LayerNorm backward kernel
    instead of autograd summing two activation-sized tensors with a separate elementwise add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, with_skip):
        be = backend_for(x)
        if be is not reference:
            why = be.layernorm_unsupported_reason(x)
            if why is not None:
                _note_fallback("layer_norm", why)
                be = reference
        x = x.contiguous()
        g32, b32 = gamma.float(), beta.float()
        y, mean, rstd = be.layernorm_fwd(x, g32, b32, eps)
        ctx.save_for_backward(x, mean, rstd, g32)
        ctx.cfg = (be, gamma.dtype, beta.dtype)
        ctx.set_materialize_grads(False)
        return y, (x.view_as(x) if with_skip else None)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, mean, rstd, g32 = ctx.saved_tensors
        be, gdt, bdt = ctx.cfg
        if dy is None:
            return dskip, None, None, None, None
        dx, dgamma, dbeta = be.layernorm_bwd(dy, x, mean, rstd, g32, dskip)
        return dx, dgamma.to(gdt), dbeta.to(bdt), None, None


def layer_norm(x, weight, bias, eps: float = 1e-5, with_skip: bool = False):
    """LayerNorm over the last dimension: one read + one write of the activation in its own dtype, fp32
    statistics (HIP kernel on bf16 GPU tensors, PyTorch fp32 oracle elsewhere).  ``with_skip=True`` returns
    ``(y, x_skip)``: use ``x_skip`` as the block's residual input (see ``_LayerNormFn``)."""
    y, skip = _LayerNormFn.apply(x, weight, bias, float(eps), bool(with_skip))
    return (y, skip) if with_skip else y


# ------------------------------------------------------------------- linear
def _lowp_weight(w, dtype):
    lp = getattr(w, "_lp", None)
    if lp is not None and lp.dtype == dtype:
        return lp
    return w if w.dtype == dtype else w.to(dtype)


def _pad_rows(t, n8):
    """zero rows up to ``n8`` (weights / bias of a head whose width is not a multiple of 8: the GEMM wants 16-B rows)"""
    if t is None or t.shape[0] == n8:
        return t
    return torch.nn.functional.pad(t, (0, 0, 0, n8 - t.shape[0]) if t.dim() == 2 else (0, n8 - t.shape[0]))


class _LinearFn(torch.autograd.Function):
    """``y = x @ w^T + b (+ residual)`` on the MFMA GEMM (csrc/kernels/gemm.hip; /root/reference/main.py:40's ``fc`` and
    every ViT projection): bias and residual ride in the epilogue; backward is one GEMM against the cached
    transposed weight, the transposing split-K weight-gradient kernel (fp32 result for the fp32 master weight) and a
    streaming column sum for the bias.  An output width that is not a multiple of 8 (a 10-class head) runs zero-padded
    to the next multiple (residual not supported there)."""

    @staticmethod
    def forward(ctx, x, w, bias, wk, residual, be):
        k = x.shape[-1]
        n = wk.shape[0]
        n8 = (n + 7) // 8 * 8
        x2 = x.reshape(-1, k)
        if n8 != n:
            assert residual is None
            wk = _pad_rows(wk, n8)
            y = be.gemm_nt(x2, wk, _pad_rows(bias, n8))[0][:, :n]
        else:
            res2 = None if residual is None else residual.reshape(-1, n)
            y, _, _ = be.gemm_nt(x2, wk, bias, res2)
        ctx.save_for_backward(x2, wk)
        ctx.cfg = (be, w.dtype, None if bias is None else bias.dtype, tuple(x.shape), residual is not None, n)
        return y.reshape(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, wk = ctx.saved_tensors
        be, wdt, bdt, x_shape, has_res, n = ctx.cfg
        dy2 = dy.reshape(-1, dy.shape[-1])
        if wk.shape[0] != n:
            dy2 = torch.nn.functional.pad(dy2, (0, wk.shape[0] - n))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = be.weight_t(wk) if wk.shape[0] == n else wk.t().contiguous()
            dx = be.gemm_nt(dy2, wt)[0].view(x_shape)
        if ctx.needs_input_grad[1]:
            dw = be.linear_wgrad(dy2, x2)[:n].to(wdt)
        if bdt is not None and ctx.needs_input_grad[2]:
            db = be.colsum(dy2.contiguous())[:n].to(bdt)
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        return dx, dw, db, None, dres, None


def linear(x, weight, bias=None, residual=None):
    """``x @ weight^T + bias (+ residual)``; ``weight`` may be an fp32 master with a bf16 working copy (``_lp``)."""
    be = backend_for(x)
    if be is not reference:
        wk = _lowp_weight(weight, x.dtype)
        why = be.linear_unsupported_reason(x, wk, padded_ok=residual is None)
        if why is None:
            return _LinearFn.apply(x, weight, bias, wk, residual, be)
        _note_fallback("linear", why)
    if weight.dtype != x.dtype or x.dtype != torch.float32:
        # low-precision compute against an fp32 master: the oracle GEMM (fp32 math, fp32 weight / bias gradients)
        return _LinearFn.apply(x, weight, bias, _lowp_weight(weight, x.dtype), residual, reference)
    y = torch.nn.functional.linear(x, weight, bias)
    return y if residual is None else y + residual


class _MlpGeluFn(torch.autograd.Function):
    """``y = gelu(x @ w0^T + b0) @ w1^T + b1 (+ residual)``: the transformer MLP as ONE node, so that no GELU or add
    kernel remains.  Forward: GELU (and the pre-activation the backward needs) in the first GEMM's epilogue, bias +
    residual in the second's.  Backward: the data gradient of the second layer is multiplied by GELU'(u) in ITS
    epilogue, which also reduces the first layer's bias gradient."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, wk0, wk1, residual, be):
        x2 = x.reshape(-1, x.shape[-1])
        h, u, _ = be.gemm_nt(x2, wk0, b0, None, None, 1, True)
        res2 = None if residual is None else residual.reshape(-1, wk1.shape[0])
        y, _, _ = be.gemm_nt(h, wk1, b1, res2)
        ctx.save_for_backward(x2, u, h, wk0, wk1)
        ctx.cfg = (be, w0.dtype, w1.dtype, None if b0 is None else b0.dtype, None if b1 is None else b1.dtype,
                   tuple(x.shape), residual is not None)
        return y.view(*x.shape[:-1], wk1.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, u, h, wk0, wk1 = ctx.saved_tensors
        be, w0dt, w1dt, b0dt, b1dt, x_shape, has_res = ctx.cfg
        dy2 = dy.reshape(-1, dy.shape[-1])
        dw1 = be.linear_wgrad(dy2, h).to(w1dt)
        db1 = be.colsum(dy2.contiguous()).to(b1dt) if b1dt is not None else None
        du, _, db0 = be.gemm_nt(dy2, be.weight_t(wk1), None, None, u, 0, False, True)
        dw0 = be.linear_wgrad(du, x2).to(w0dt)
        dx = be.gemm_nt(du, be.weight_t(wk0))[0].view(x_shape) if ctx.needs_input_grad[0] else None
        db0 = db0.to(b0dt) if b0dt is not None else None
        dres = dy if (has_res and ctx.needs_input_grad[7]) else None
        return dx, dw0, db0, dw1, db1, None, None, dres, None


def mlp_gelu(x, w0, b0, w1, b1, residual=None):
    be = backend_for(x)
    wk0, wk1 = _lowp_weight(w0, x.dtype), _lowp_weight(w1, x.dtype)
    if be is not reference:
        why = be.linear_unsupported_reason(x, wk0) or be.linear_unsupported_reason(x, wk1)
        if why is not None:
            _note_fallback("mlp_gelu", why)
            be = reference
    if be is reference and x.dtype == torch.float32 and w0.dtype == torch.float32:
        y = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(x, w0, b0)), w1, b1)
        return y if residual is None else y + residual
    return _MlpGeluFn.apply(x, w0, b0, w1, b1, wk0, wk1, residual, be)


# ----------------------------------------------------------------- vit glue
class _PatchEmbedFn(torch.autograd.Function):
    """A ``kernel = stride = p`` convolution of the image (ViT ``conv_proj``) as patch extraction + ONE GEMM.  Patches
    and weight both use the (py, px, c) element order, which is the physical order of the channels_last weight and
    of p*3 contiguous image elements, so neither operand is re-laid-out."""

    @staticmethod
    def forward(ctx, x, w, bias, wk2, p, be):
        b = x.shape[0]
        patches = be.patchify(x, p)
        y, _, _ = be.gemm_nt(patches, wk2, bias)
        ctx.save_for_backward(patches)
        ctx.cfg = (be, tuple(w.shape), w.dtype, None if bias is None else bias.dtype)
        return y.view(b, -1, wk2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        be, w_shape, wdt, bdt = ctx.cfg
        d, c, p, _ = w_shape
        dy2 = dy.reshape(-1, d)
        dw = db = None
        if ctx.needs_input_grad[1]:
            dw = be.linear_wgrad(dy2, patches).to(wdt).view(d, p, p, c).permute(0, 3, 1, 2)
        if bdt is not None and ctx.needs_input_grad[2]:
            db = be.colsum(dy2.contiguous()).to(bdt)
        return None, dw, db, None, None, None


def patch_embed(x, weight, bias, p: int):
    """``conv2d(x, weight, bias, stride=p)`` for ``kernel_size == p`` -> tokens ``[B, (S/p)^2, D]``"""
    be = backend_for(x)
    d = weight.shape[0]
    if be is not reference:
        wk = _lowp_weight(weight, x.dtype)
        wk2 = be.as_matrix_param(wk, d)
        why = be.patchify_unsupported_reason(x, p) or be.linear_unsupported_reason(x, wk2)
        if why is None:
            return _PatchEmbedFn.apply(x, weight, bias, wk2, p, be)
        _note_fallback("patch_embed", why)
    if x.dtype != weight.dtype:
        wk = _lowp_weight(weight, x.dtype)
        return _PatchEmbedFn.apply(x, weight, bias, reference.as_matrix_param(wk, d), p, reference)
    y = torch.nn.functional.conv2d(x, weight, bias, stride=p)
    return y.flatten(2).transpose(1, 2)


class _VitEmbedFn(torch.autograd.Function):
    """``cat(class_token, tokens) + pos_embedding`` in one pass (and one pass backward: token gradient copy + the
    batch sums that are the position-embedding / class-token gradients)."""

    @staticmethod
    def forward(ctx, tokens, cls, pos, be):
        ctx.cfg = (be, cls.dtype, pos.dtype, tuple(cls.shape), tuple(pos.shape))
        return be.vit_embed_fwd(tokens, cls, pos)

    @staticmethod
    def backward(ctx, dx):
        be, cdt, pdt_, cshape, pshape = ctx.cfg
        dtok, dpos = be.vit_embed_bwd(dx)
        return (dtok if ctx.needs_input_grad[0] else None, dpos[0].to(cdt).view(cshape), dpos.to(pdt_).view(pshape), None)


def vit_embed(tokens, class_token, pos_embedding):
    be = backend_for(tokens)
    if be is not reference and (tokens.dtype != torch.bfloat16 or tokens.shape[-1] % 8):
        _note_fallback("vit_embed", f"{tokens.dtype} width {tokens.shape[-1]}")
        be = reference
    if be is reference and tokens.dtype == class_token.dtype:
        b = tokens.shape[0]
        return torch.cat([class_token.expand(b, -1, -1), tokens], dim=1) + pos_embedding
    return _VitEmbedFn.apply(tokens, class_token, pos_embedding, be)


# -------------------------------------------------------------------- pools
class _MaxPool3x3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        be = backend_for(x)
        x = _cl(x)
        if be is not reference and be.bn_unsupported_reason(x) is not None:
            _note_fallback("maxpool", "dtype/layout")
            be = reference
        y, aux = be.maxpool3x3s2_fwd(x)
        ctx.save_for_backward(x, y)
        ctx.be, ctx.aux = be, aux
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        return ctx.be.maxpool3x3s2_bwd(_cl(dy), x, y, ctx.aux)


def maxpool3x3s2(x):
    return _MaxPool3x3s2Fn.apply(x)


class _GlobalAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        be = backend_for(x)
        x = _cl(x)
        if be is not reference and be.bn_unsupported_reason(x) is not None:
            _note_fallback("avgpool", "dtype/layout")
            be = reference
        ctx.be, ctx.shape = be, tuple(x.shape)
        return be.global_avgpool_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ctx.be.global_avgpool_bwd(dy.contiguous(), ctx.shape)


class _BNReluPoolFn(torch.autograd.Function):
    """``maxpool3x3/2/1(relu(bn(u)))`` as ONE node (the ResNet stem tail: /root/reference/main.py:40 builds
    torchvision's conv1 -> bn1 -> relu -> maxpool).  The forward reads the conv output once and writes the pooled
    tensor, the winning taps and the 1-bit ReLU mask; the full-resolution activation is never written.  Backward:
    pool backward, then the BN backward kernels gated by the mask."""

    @staticmethod
    def forward(ctx, u, gamma, beta, stats, cfg: BNConfig):
        be = backend_for(u)
        u = _cl(u)
        if be is not reference:
            why = be.bn_unsupported_reason(u)
            if why is not None:
                _note_fallback("bn_relu_maxpool", why)
                be = reference
        mean, invstd, count, count_t, world, ctx.ipc = _bn_moments(be, u, stats, cfg)
        c = u.shape[1]
        g32 = gamma.float() if gamma is not None else torch.ones(c, device=u.device)
        b32 = beta.float() if beta is not None else torch.zeros(c, device=u.device)
        y, aux, act = be.bn_relu_maxpool_fwd(u, mean, invstd, g32, b32)
        ctx.save_for_backward(u, y, aux, act, mean, invstd, g32, count_t)
        ctx.cfg = (be, cfg, world, count, None if gamma is None else gamma.dtype, None if beta is None else beta.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, y, aux, act, mean, invstd, g32, count_t = ctx.saved_tensors
        be, cfg, world, count, gdt, bdt = ctx.cfg
        if (_FUSE_STEM_POOL_BWD and be is not reference and cfg.training and (world == 1 or ctx.ipc is not None)
                and be.bn_relu_maxpool_bwd_ok(u)):
            # both BN-backward passes gather the pooled gradient themselves: no full-resolution dz in memory
            du, sums = be.bn_relu_maxpool_bwd(_cl(dy), aux, u, act, mean, invstd, g32, count_t if world > 1 else count,
                                              ipc=ctx.ipc.native if world > 1 else None)
            dgamma = sums[1].to(gdt) if (gdt is not None and ctx.needs_input_grad[1]) else None
            dbeta = sums[0].to(bdt) if (bdt is not None and ctx.needs_input_grad[2]) else None
            return (du if ctx.needs_input_grad[0] else None), dgamma, dbeta, None, None
        # (the oracle's pool backward recomputes the pooling from its input: hand it the activation it kept)
        dz = be.maxpool3x3s2_bwd(_cl(dy), act if be is reference else u, y, aux)
        psums = be.bn_bwd_reduce(dz, act, u, mean, invstd, True)
        if cfg.training and world > 1 and ctx.ipc is not None:
            du, _, sums = be.bn_bwd_elemt(dz, act, u, mean, invstd, g32, psums, count_t, True, False, ipc=ctx.ipc.native)
        elif cfg.training and world > 1:
            sums = be.bn_sum_partials(psums)
            gsums = sums.clone()
            _small_all_reduce(gsums, cfg.group)
            du, _, _ = be.bn_bwd_elemt(dz, act, u, mean, invstd, g32, gsums.view(1, 2, -1), count_t, True, False)
        elif cfg.training:
            du, _, sums = be.bn_bwd_elemt(dz, act, u, mean, invstd, g32, psums, count, True, False)
        else:
            sums = be.bn_sum_partials(psums)
            du, _, _ = be.bn_bwd_elemt(dz, act, u, mean, invstd, g32, torch.zeros_like(sums).view(1, 2, -1), 1.0,
                                       True, False)
        dgamma = sums[1].to(gdt) if (gdt is not None and ctx.needs_input_grad[1]) else None
        dbeta = sums[0].to(bdt) if (bdt is not None and ctx.needs_input_grad[2]) else None
        return (du if ctx.needs_input_grad[0] else None), dgamma, dbeta, None, None


def bn_relu_maxpool(u, gamma, beta, cfg: BNConfig, stats=None):
    return _BNReluPoolFn.apply(u, gamma, beta, stats, cfg)


def global_avgpool(x):
    """``[B,C,H,W] -> [B,C]`` (adaptive avg-pool to 1x1 + flatten)."""
    return _GlobalAvgPoolFn.apply(x)


# --------------------------------------------------------------------- loss
class _SoftmaxCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        be = backend_for(logits)
        loss, dlogits = be.softmax_ce_fwd_bwd(logits.contiguous(), labels.contiguous())
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        return dlogits * dloss.to(dlogits.dtype), None


def softmax_cross_entropy(logits, labels):
    """Mean-reduced CE; forward already produces ``(softmax - onehot)/B`` (N11)."""
    return _SoftmaxCEFn.apply(logits, labels)


# ------------------------------------------------------------------ helpers
def conv_bn_act(x, conv, bn, relu: bool = True, residual=None, return_skip: bool = False,
                lazy_residual_grad: bool = False, defer_apply: bool = False, compact_skip_grad: bool = False):
    """``act(bn(conv(x)) [+ residual])`` through the fused nodes.

    ``conv``/``bn`` are modules (parameter containers); bias-free square convs with dilation 1 and
    groups 1 take the fused path, anything else falls back to the module's own forward.
    ``return_skip=True`` also returns ``x`` routed through the conv node (use it as the block's
    identity branch so its gradient is fused into this conv's data-gradient kernel).
    """
    from .modules import BatchNorm2d, Conv2d  # local: avoid cycle

    def done(out, skip):
        return (out, skip) if return_skip else out

    if _PLAIN_TORCH:
        # calibration mode: the same network through stock PyTorch-ROCm ops (MIOpen conv, ATen BN)
        # (an autograd-visible cast of the fp32 master weight, as autocast does: the detached low-precision shadow
        # the native nodes read would leave the convolution weights without a gradient)
        y = torch.nn.functional.conv2d(x, conv.weight.to(x.dtype), None, conv.stride, conv.padding)
        y = torch.nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training,
                                           bn.momentum if bn.momentum is not None else 0.1, bn.eps)
        if residual is not None:
            y = y + residual
        return done(torch.relu(y) if relu else y, x)

    fusable = (
        isinstance(conv, Conv2d) and isinstance(bn, BatchNorm2d) and conv.bias is None and conv.groups == 1
        and conv.dilation == (1, 1) and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
        and conv.kernel_size[0] == conv.kernel_size[1] and isinstance(conv.padding, tuple)
    )
    if not fusable:
        y = bn(conv(x))
        if residual is not None:
            y = y + residual
        return done(torch.relu(y) if relu else y, x)
    want_stats = bn.training or not bn.track_running_stats
    skip = x
    if return_skip and torch.is_grad_enabled() and x.requires_grad:
        y, stats, skip = conv2d(x, conv.weight, conv.stride[0], conv.padding[0], want_stats=want_stats, with_skip=True)
    else:
        y, stats = conv2d(x, conv.weight, conv.stride[0], conv.padding[0], want_stats=want_stats,
                          compact_skip_grad=compact_skip_grad)
    return done(bn(y, residual=residual, relu=relu, stats=stats, lazy_residual_grad=lazy_residual_grad,
                   defer_apply=defer_apply), skip)


def conv_bn_relu_maxpool(x, conv, bn, pool):
    """``pool(relu(bn(conv(x))))`` for the ResNet stem; the BN + ReLU + 3x3/2/1 max-pool tail runs as one fused
    node when the modules are the plain ones, else the unfused composition."""
    from .modules import BatchNorm2d, Conv2d  # local: avoid cycle

    from .modules import MaxPool3x3s2

    std_pool = isinstance(pool, MaxPool3x3s2) or (
        isinstance(pool, tnn.MaxPool2d) and pool.kernel_size in (3, (3, 3)) and pool.stride in (2, (2, 2))
        and pool.padding in (1, (1, 1)) and pool.dilation in (1, (1, 1)) and not pool.ceil_mode
        and not pool.return_indices)
    fusable = (
        std_pool and _FUSE_STEM_POOL and not _PLAIN_TORCH and isinstance(conv, Conv2d) and isinstance(bn, BatchNorm2d)
        and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1) and conv.stride[0] == conv.stride[1]
        and conv.padding[0] == conv.padding[1] and conv.kernel_size[0] == conv.kernel_size[1]
    )
    if not fusable:
        y = conv_bn_act(x, conv, bn, relu=True)
        return maxpool3x3s2(y) if std_pool else pool(y)
    want_stats = bn.training or not bn.track_running_stats
    y, stats = conv2d(x, conv.weight, conv.stride[0], conv.padding[0], want_stats=want_stats)
    cfg = bn._cfg()
    return bn_relu_maxpool(y, bn.weight, bn.bias, cfg, stats if cfg.training else None)


_FUSE_STEM_POOL = os.environ.get("PDT_FUSE_STEM_POOL", "1") != "0"  # A/B knob
