"""state_dict checkpoints: save on rank 0, resume on every rank.

The reference never saves anything (its last live statement is the
``TrainTime`` footer, ``/root/reference/main.py:117``); the north star asks for
the "state_dict checkpoint format".  It is what a user of that script would
write::

    {"model": net.module.state_dict(),      # un-prefixed torchvision keys, fp32
     "optimizer": opt.state_dict(),         # stock torch.optim layout
     "epoch": e, "global_step": s, "sampler": {...}, "extra": {...}}

Model tensors are saved as plain contiguous CPU fp32 (no arena views, no
``module.`` prefix) so the file loads with ``strict=True`` into a stock
torchvision ResNet and vice versa.  Loading copies INTO the flat arena.
Files are written atomically (tmp + rename) and read with
``weights_only=True`` (no code execution from checkpoint files).
"""
from __future__ import annotations

import os
import tempfile

import torch
import torch.nn as tnn

from .. import dist as pdist

FORMAT_VERSION = 1


def unwrap(model: tnn.Module) -> tnn.Module:
    return model.module if hasattr(model, "module") and isinstance(model.module, tnn.Module) else model


def model_state_dict(model: tnn.Module) -> dict:
    sd = unwrap(model).state_dict()
    return {k: v.detach().to("cpu").contiguous().clone() for k, v in sd.items()}


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().to("cpu").contiguous().clone()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def save_checkpoint(path: str, model: tnn.Module, optimizer=None, epoch: int = 0, global_step: int = 0,
                    sampler=None, extra: dict | None = None, rank0_only: bool = True) -> bool:
    """Write the checkpoint (rank 0 only by default). Returns True when this rank wrote."""
    if rank0_only and pdist.get_rank() != 0:
        return False
    payload = {
        "format_version": FORMAT_VERSION,
        "model": model_state_dict(model),
        "optimizer": _to_cpu(optimizer.state_dict()) if optimizer is not None else None,
        "epoch": int(epoch),
        "global_step": int(global_step),
        "sampler": sampler.state_dict() if (sampler is not None and hasattr(sampler, "state_dict")) else None,
        "extra": extra or {},
    }
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    fd, tmp = tempfile.mkstemp(dir=d, prefix=".ckpt-", suffix=".tmp")
    os.close(fd)
    try:
        torch.save(payload, tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return True


def load_checkpoint(path: str, model: tnn.Module, optimizer=None, sampler=None, strict: bool = True,
                    map_location="cpu") -> dict:
    """Restore model (+ optimizer, sampler). Call BEFORE wrapping in DDP so the
    initial broadcast propagates rank 0's state, or after (every rank loads the
    same file). Returns ``{"epoch", "global_step", "extra"}``."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    target = unwrap(model)
    target.load_state_dict(ckpt["model"], strict=strict)
    arena = getattr(model, "param_arena", None)
    if arena is not None:
        arena.refresh_lowp()
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    if sampler is not None and ckpt.get("sampler") is not None and hasattr(sampler, "load_state_dict"):
        sampler.load_state_dict(ckpt["sampler"])
    return {"epoch": ckpt.get("epoch", 0), "global_step": ckpt.get("global_step", 0), "extra": ckpt.get("extra", {})}
