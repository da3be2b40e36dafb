"""Auxiliary subsystems: checkpoint, profiling, metrics, extension loading."""
from . import checkpoint, env, metrics, profiling  # noqa: F401

__all__ = ["checkpoint", "env", "metrics", "profiling"]
