"""Locating and loading the native (C++/HIP) extension.

The extension is built IN-TREE by ``__graft_entry__.build()`` (or
``python -m pytorch_distributed_training_amd.csrc.build``) into
``pytorch_distributed_training_amd/_C*.so``.  There is no JIT and no silent
fallback on a GPU box: if a tensor lives on the GPU and the extension is
missing, ops raise (``require_native``).
"""
from __future__ import annotations

import importlib
import os
import threading

_lock = threading.Lock()
_ext = None
_ext_err: Exception | None = None
_tried = False


def _load():
    global _ext, _ext_err, _tried
    with _lock:
        if _tried:
            return _ext
        _tried = True
        if os.environ.get("PDT_DISABLE_NATIVE", "0") == "1":
            _ext_err = RuntimeError("native extension disabled by PDT_DISABLE_NATIVE=1")
            return None
        try:
            import torch  # noqa: F401  (libtorch must be loaded before the extension)

            _ext = importlib.import_module("pytorch_distributed_training_amd._C")
        except Exception as e:  # pragma: no cover - depends on build state
            _ext_err = e
            _ext = None
        return _ext


def native_available() -> bool:
    """True when the in-tree ``_C`` extension imports."""
    return _load() is not None


def native():
    """Return the extension module or ``None``."""
    return _load()


def require_native():
    """Return the extension module; raise loudly when it is missing."""
    ext = _load()
    if ext is None:
        raise RuntimeError(
            "pytorch_distributed_training_amd._C is not built/loadable "
            f"({_ext_err!r}). Run `python __graft_entry__.py build` (hipcc, gfx950)."
        )
    return ext


def gpu_native_ok() -> bool:
    """True when HIP kernels can actually be launched in this process."""
    import torch

    return torch.cuda.is_available() and native_available()
