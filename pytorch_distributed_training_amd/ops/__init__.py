"""Op layer: ``native`` = hand-written gfx950 HIP kernels, ``reference`` = PyTorch oracles.

``backend_for(t)`` picks the implementation for a tensor: GPU tensors run the
HIP kernels (and raise if the extension is missing: there is no silent eager
fallback on a GPU box), CPU tensors run the oracles (CPU/gloo plumbing).
"""
from __future__ import annotations

import os

from . import reference
from ..utils.env import require_native

_FORCE_REFERENCE = os.environ.get("PDT_FORCE_REFERENCE_OPS", "0") == "1"


def backend_for(t):
    if t.is_cuda and not _FORCE_REFERENCE:
        require_native()
        from . import native

        return native
    return reference


def set_force_reference(flag: bool) -> None:
    """Testing hook: run the PyTorch oracles even for GPU tensors."""
    global _FORCE_REFERENCE
    _FORCE_REFERENCE = bool(flag)


def force_reference() -> bool:
    return _FORCE_REFERENCE


__all__ = ["backend_for", "reference", "set_force_reference", "force_reference"]
