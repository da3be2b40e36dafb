"""Gradient-bucket planning sized for xGMI.

Stock DDP: 25 MiB cap, 1 MiB first bucket, buckets assigned in *reverse*
parameter order, rebuilt from the observed gradient-ready order after the
first iteration.  xGMI is 7 point-to-point links per GPU, so
each link carries ``bucket/8`` per phase of a direct all-reduce; buckets of
8-32 MiB keep per-link transfers past the latency knee while preserving overlap
granularity, and the last-produced bucket (conv1/bn1) cannot overlap with
anything, so it is kept small.  Defaults: first bucket 1 MiB,
cap 16 MiB; ``autotune_bucket_cap`` times 4/8/16/32 MiB all-reduces on the live
communicator and picks the best bus bandwidth.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import torch

DEFAULT_BUCKET_CAP_MB = 16.0
DEFAULT_FIRST_BUCKET_MB = 1.0
_ALIGN_ELEMS = 128


@dataclass
class BucketSpec:
    param_indices: list[int] = field(default_factory=list)
    offsets: list[int] = field(default_factory=list)  # element offsets inside the flat bucket
    numel: int = 0  # padded total


def _pad(n: int) -> int:
    return (n + _ALIGN_ELEMS - 1) // _ALIGN_ELEMS * _ALIGN_ELEMS


def plan_buckets(numels: list[int], order: list[int], elem_size: int, cap_bytes: int,
                 first_cap_bytes: int | None = None) -> list[BucketSpec]:
    """Greedy fill in ``order`` (a permutation of parameter indices).  A bucket
    closes when adding the next tensor would exceed its cap (a single tensor
    larger than the cap gets a bucket of its own)."""
    buckets: list[BucketSpec] = []
    cur = BucketSpec()
    cap = first_cap_bytes if first_cap_bytes is not None else cap_bytes
    for idx in order:
        n = _pad(numels[idx])
        if cur.param_indices and (cur.numel + n) * elem_size > cap:
            buckets.append(cur)
            cur = BucketSpec()
            cap = cap_bytes
        cur.param_indices.append(idx)
        cur.offsets.append(cur.numel)
        cur.numel += n
    if cur.param_indices:
        buckets.append(cur)
    return buckets


def initial_order(num_params: int) -> list[int]:
    """Reverse registration order approximates gradient-ready order (fc first)."""
    return list(range(num_params - 1, -1, -1))


def order_hash(order: list[int]) -> int:
    h = 1469598103934665603
    for v in order:
        h = ((h ^ (v + 1)) * 1099511628211) & 0x7FFFFFFFFFFFFFFF
    return h


def autotune_bucket_cap(all_reduce_fn, device, dtype=torch.float32, candidates_mb=(4, 8, 16, 32), iters: int = 5):
    """Time an all-reduce of each candidate size; return (best_mb, {mb: seconds}).
    ``all_reduce_fn(tensor)`` must block until the collective is complete."""
    results = {}
    for mb in candidates_mb:
        n = int(mb * (1 << 20)) // torch.empty((), dtype=dtype).element_size()
        t = torch.zeros(n, dtype=dtype, device=device)
        all_reduce_fn(t)  # warm-up (communicator / channel setup)
        t0 = time.perf_counter()
        for _ in range(iters):
            all_reduce_fn(t)
        results[mb] = (time.perf_counter() - t0) / iters
    best = max(results, key=lambda mb: mb / results[mb])  # highest bytes/s
    return best, results
