"""Data parallelism: arenas, bucket planning, reducers, the DDP wrapper (``main.py:83``)."""
from .arena import BufferArena, ParamArena
from .buckets import BucketSpec, autotune_bucket_cap, initial_order, plan_buckets
from .comm import TorchComm
from .ddp import DistributedDataParallel
from .reducer_py import PyReducer

__all__ = ["DistributedDataParallel", "ParamArena", "BufferArena", "PyReducer", "TorchComm", "BucketSpec",
           "plan_buckets", "initial_order", "autotune_bucket_cap"]
