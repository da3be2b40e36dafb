"""DistributedSampler with the index math of the stock sampler.

The reference shards CIFAR-100 with ``DistributedSampler(trainset)`` and calls
``sampler.set_epoch(e)`` each epoch (``/root/reference/main.py:53,93``).  For
the same ``(len, world, rank, seed, epoch, shuffle, drop_last)`` this class
yields exactly the indices ``torch.utils.data.DistributedSampler`` yields, so a
run here visits samples in the same order as a run of the reference
(parity is a unit test).

Addition: ``state_dict``/``load_state_dict`` and ``iter_from(start)`` for
mid-epoch resume.
"""
from __future__ import annotations

import math
from typing import Iterator, Sized

import torch
import torch.distributed as tdist


class DistributedSampler:
    def __init__(self, dataset: Sized, num_replicas: int | None = None, rank: int | None = None,
                 shuffle: bool = True, seed: int = 0, drop_last: bool = False):
        if num_replicas is None:
            num_replicas = tdist.get_world_size() if (tdist.is_available() and tdist.is_initialized()) else 1
        if rank is None:
            rank = tdist.get_rank() if (tdist.is_available() and tdist.is_initialized()) else 0
        if not 0 <= rank < num_replicas:
            raise ValueError(f"invalid rank {rank}: must be in [0, {num_replicas})")
        self.dataset = dataset
        self.num_replicas = int(num_replicas)
        self.rank = int(rank)
        self.epoch = 0
        self.shuffle = shuffle
        self.seed = int(seed)
        self.drop_last = drop_last
        n = len(dataset)
        if drop_last and n % self.num_replicas != 0:
            # the tail that does not divide evenly is dropped
            self.num_samples = math.ceil((n - self.num_replicas) / self.num_replicas)
        else:
            self.num_samples = math.ceil(n / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def global_indices(self) -> list[int]:
        """The padded / truncated global order for the current epoch."""
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            indices = torch.randperm(n, generator=g).tolist()
        else:
            indices = list(range(n))
        if not self.drop_last:
            pad = self.total_size - len(indices)
            if pad <= len(indices):
                indices += indices[:pad]
            else:
                indices += (indices * math.ceil(pad / len(indices)))[:pad]
        else:
            indices = indices[: self.total_size]
        assert len(indices) == self.total_size
        return indices

    def local_indices(self) -> list[int]:
        idx = self.global_indices()[self.rank: self.total_size: self.num_replicas]
        assert len(idx) == self.num_samples
        return idx

    def __iter__(self) -> Iterator[int]:
        return iter(self.local_indices())

    def iter_from(self, start: int) -> Iterator[int]:
        """Resume inside an epoch after ``start`` samples of this rank."""
        return iter(self.local_indices()[start:])

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def state_dict(self) -> dict:
        return {"epoch": self.epoch, "seed": self.seed}

    def load_state_dict(self, state: dict) -> None:
        self.epoch = int(state.get("epoch", 0))
        self.seed = int(state.get("seed", self.seed))
