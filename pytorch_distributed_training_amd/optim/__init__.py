"""Fused optimizers (reference ``main.py:80``: Adam, lr 1e-3)."""
from .fused import SGD, Adam, AdamW, build_optimizer

__all__ = ["Adam", "AdamW", "SGD", "build_optimizer"]
