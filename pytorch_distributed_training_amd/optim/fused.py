"""Fused Adam / AdamW / SGD-momentum over flat buffers.

The reference uses ``torch.optim.Adam(net.parameters(), lr=0.001)`` built
*before* ``.cuda()``, SyncBN conversion and DDP (``/root/reference/main.py:80-83``)
and steps every iteration (``main.py:104``); stock Adam on GPU is ~10
multi-tensor launches per step.

Here one HIP launch per ≤64 tensors updates ``p, m, v`` (fp32 master state),
applies weight decay / bias correction, and writes the bf16 working copy the
MFMA kernels read (``param._lp``) in the same pass; gradients are read straight
from the reducer's flat buckets (``.grad`` aliases them).  State is allocated
lazily at the first ``step()``, so constructing the optimizer on CPU parameters
that later move to the GPU and into a flat arena works like the reference
order requires.  ``state_dict()`` has the stock layout (``state[idx] =
{step, exp_avg, exp_avg_sq}`` / ``{momentum_buffer}``), so checkpoints
interchange with ``torch.optim``.
"""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from ..ops import backend_for, reference


def _flat_state_like(params, store: dict, key: str):
    """Allocate optimizer state with the same flat layout as the parameter arena
    when the parameters are arena-backed; per-tensor otherwise."""
    arena_backed = all(hasattr(p, "_arena_slot") for p in params)
    if arena_backed and params:
        lo = min(p._arena_slot[1] for p in params)
        hi = max(p._arena_slot[1] + p._arena_slot[2] for p in params)
        flat = torch.zeros(hi - lo, dtype=torch.float32, device=params[0].device)
        store[key] = flat
        out = []
        for p in params:
            _, off, n = p._arena_slot
            seg = flat[off - lo: off - lo + n]
            out.append(torch.as_strided(seg, p.shape, p.stride()) if p.dim() > 0 else seg.view(()))
        return out
    return [torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format) for p in params]


def _grad_like(p: torch.Tensor) -> torch.Tensor:
    g = p.grad
    if g.is_sparse:
        raise RuntimeError("sparse gradients are not supported")
    if g.stride() != p.stride():
        g = torch.empty_like(p, dtype=g.dtype).copy_(g)  # same physical order as the parameter
    return g


class _FusedBase(Optimizer):
    _state_keys: tuple[str, ...] = ()

    def _collect(self, group):
        return [p for p in group["params"] if p.grad is not None]

    def _lazy_state(self, group, params):
        missing = [p for p in params if not all(k in self.state[p] for k in self._state_keys)]
        if not missing:
            return
        flats = self.__dict__.setdefault("_flats", {})  # keeps the flat state arenas alive
        for key in self._state_keys:
            for p, t in zip(missing, _flat_state_like(missing, flats, f"{key}:{len(flats)}")):
                self.state[p][key] = t
        for p in missing:
            self.state[p].setdefault("step", torch.tensor(0.0))

    # -- step counters: one Python int per group on the hot path; the per-parameter ``state[p]["step"]``
    #    tensors of the stock layout are only materialised for state_dict() ---------------------------------
    def _group_step(self, gi: int, params) -> int | None:
        """Current step count of group ``gi`` (None if its parameters disagree, e.g. after a partial load)."""
        steps = self.__dict__.setdefault("_fused_steps", {})
        if gi not in steps:
            seen = {int(float(self.state[p]["step"])) for p in params if "step" in self.state[p]}
            if len(seen) > 1:
                return None
            steps[gi] = seen.pop() if seen else 0
        return steps[gi]

    def _sync_step_tensors(self) -> None:
        steps = self.__dict__.get("_fused_steps", {})
        devs = self.__dict__.get("_step_dev", {})
        for gi, group in enumerate(self.param_groups):
            if gi in devs and gi in steps:  # graph replays advance only the device counter
                steps[gi] = int(round(float(devs[gi].item())))
            if gi not in steps:
                continue
            for p in group["params"]:
                if p in self.state and "step" in self.state[p]:
                    self.state[p]["step"] = torch.tensor(float(steps[gi]))

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__["_state_loaded"] = True
        self.__dict__.pop("_fused_steps", None)  # re-derived from the loaded per-parameter steps
        for gi, t in self.__dict__.get("_step_dev", {}).items():
            params = [p for p in self.param_groups[gi]["params"] if p in self.state]
            st = self._group_step(gi, params)
            if st is not None:
                t.fill_(float(st))

    def set_capturable(self, on: bool = True) -> None:
        """Keep the step count in device memory so a captured hipGraph replays with the right bias
        correction (the host only advances it when Python actually runs ``step()``)."""
        self.__dict__["_capturable"] = bool(on)

    def _move_loaded_state(self, params):
        # after load_state_dict the state tensors may sit on another device / layout (checked once per load)
        if not self.__dict__.get("_state_loaded", False):
            return
        self.__dict__["_state_loaded"] = False
        for p in params:
            st = self.state[p]
            for k in self._state_keys:
                t = st.get(k)
                if t is not None and (t.device != p.device or t.stride() != p.stride() or t.dtype != torch.float32):
                    st[k] = torch.empty_like(p, dtype=torch.float32).copy_(t)


class Adam(_FusedBase):
    _state_keys = ("exp_avg", "exp_avg_sq")
    _adamw = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale: float = 1.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = grad_scale

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = self._collect(group)
            if not params:
                continue
            self._lazy_state(group, params)
            self._move_loaded_state(params)
            b1, b2 = group["betas"]
            step = self._group_step(gi, params)
            grads = [_grad_like(p) for p in params]
            ms = [self.state[p]["exp_avg"] for p in params]
            vs = [self.state[p]["exp_avg_sq"] for p in params]
            lps = [getattr(p, "_lp", None) for p in params]
            be = backend_for(params[0])
            if be is not reference and step is not None:
                step += 1
                self._fused_steps[gi] = step
                step_dev = None
                if self.__dict__.get("_capturable", False):
                    devs = self.__dict__.setdefault("_step_dev", {})
                    if gi not in devs:
                        devs[gi] = torch.full((1,), float(step - 1), dtype=torch.float32, device=params[0].device)
                    step_dev = devs[gi]
                    step_dev.add_(1.0)
                be.multi_adam(params, grads, ms, vs, lps, group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], step, self._adamw, self.grad_scale, step_dev)
            else:
                self._sync_step_tensors()
                self.__dict__.get("_fused_steps", {}).pop(gi, None)
                for p, g, m, v, lp in zip(params, grads, ms, vs, lps):
                    self.state[p]["step"] += 1
                    reference.adam_step(p, g, m, v, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                        float(self.state[p]["step"]), self._adamw, self.grad_scale, lp)
        return loss


class AdamW(Adam):
    _adamw = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale: float = 1.0):
        super().__init__(params, lr, betas, eps, weight_decay, grad_scale)


class SGD(_FusedBase):
    _state_keys = ("momentum_buffer",)

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False,
                 grad_scale: float = 1.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))
        self.grad_scale = grad_scale

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = self._collect(group)
            if not params:
                continue
            first = [not all(k in self.state[p] for k in self._state_keys) for p in params]
            self._lazy_state(group, params)
            self._move_loaded_state(params)
            step = self._group_step(gi, params)
            if step is not None:
                self._fused_steps[gi] = step + 1
            else:
                for p in params:
                    self.state[p]["step"] += 1
            grads = [_grad_like(p) for p in params]
            bufs = [self.state[p]["momentum_buffer"] for p in params]
            lps = [getattr(p, "_lp", None) for p in params]
            be = backend_for(params[0])
            uniform_first = all(first) or not any(first)
            if be is not reference and uniform_first:
                be.multi_sgd(params, grads, bufs, lps, group["lr"], group["momentum"], group["dampening"],
                             group["weight_decay"], group["nesterov"], all(first), self.grad_scale)
            else:
                for p, g, buf, lp, f in zip(params, grads, bufs, lps, first):
                    reference.sgd_step(p, g, buf, group["lr"], group["momentum"], group["dampening"],
                                       group["weight_decay"], group["nesterov"], f, self.grad_scale, lp)
        return loss


def build_optimizer(name: str, params, lr: float, momentum: float = 0.9, weight_decay: float = 0.0):
    name = name.lower()
    if name == "adam":
        return Adam(params, lr=lr, weight_decay=weight_decay)
    if name == "adamw":
        return AdamW(params, lr=lr, weight_decay=weight_decay if weight_decay else 1e-2)
    if name == "sgd":
        return SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
    raise ValueError(f"unknown optimizer {name!r}")
