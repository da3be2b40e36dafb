"""Data path: sampler, datasets, loader (reference ``main.py:42-63``)."""
from .datasets import CIFAR100Binary, SyntheticCIFAR, SyntheticImageNet, build_dataset
from .loader import DataLoader, SyntheticDeviceLoader, default_collate
from .sampler import DistributedSampler

__all__ = [
    "DistributedSampler", "DataLoader", "SyntheticDeviceLoader", "default_collate",
    "SyntheticCIFAR", "SyntheticImageNet", "CIFAR100Binary", "build_dataset",
]
