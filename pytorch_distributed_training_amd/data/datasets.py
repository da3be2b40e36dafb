"""Datasets: CIFAR-shaped and ImageNet-shaped synthetic data, optional real CIFAR-100.

The reference trains on ``CIFAR100(root='dataset', download=True,
transform=ToTensor())`` (``/root/reference/main.py:42-51``): 50 000 train /
10 000 test RGB 32x32 uint8 images, converted per sample to float32 CHW in
[0, 1] with int labels in [0, 100).  There is no network in this environment,
so the default datasets are synthetic with the same shapes / dtypes / lengths
(SURVEY §2.2 C5); machines that have the files get the real set from either
distribution: ``cifar-100-binary/{train,test}.bin`` (plain records) or
``cifar-100-python/{train,test}`` -- the tree the reference's
``download=True`` leaves in ``dataset/`` -- read by a loader that executes
nothing from the file (``_load_cifar_python``).

Every dataset implements ``__getitem__`` (ToTensor semantics) and the faster
``get_batch(indices)`` used by our loader (one gather instead of B Python
calls; the reference's per-sample PIL path is CPU-bound).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch


class _ArrayImageDataset:
    """uint8 HWC image pool + int64 labels, ToTensor on access."""

    def __init__(self, images_u8: torch.Tensor, labels: torch.Tensor, length: int | None = None):
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4  # [N, H, W, C]
        self.images = images_u8
        self.labels = labels.to(torch.int64)
        self._pool = images_u8.shape[0]
        self._len = int(length if length is not None else self._pool)

    def __len__(self) -> int:
        return self._len

    @property
    def sample_shape(self) -> tuple[int, int, int]:
        n, h, w, c = self.images.shape
        return (c, h, w)

    def __getitem__(self, i: int):
        if not -self._len <= i < self._len:
            raise IndexError(i)
        j = (i % self._len) % self._pool
        img = self.images[j].permute(2, 0, 1).to(torch.float32).div_(255.0)
        return img, int(self.labels[j])

    def get_batch(self, indices, out: torch.Tensor | None = None, out_labels: torch.Tensor | None = None,
                  raw_u8: bool = False):
        """Gather a batch.  ``raw_u8=True`` returns the uint8 NHWC block (the
        device-side cast kernel then does /255 + layout + bf16 in one pass)."""
        idx = torch.as_tensor(indices, dtype=torch.int64) % self._pool
        labels = self.labels.index_select(0, idx)
        block = self.images.index_select(0, idx)  # [B, H, W, C] uint8
        if raw_u8:
            if out is not None:
                out[: block.shape[0]].copy_(block)
                block = out[: block.shape[0]]
        else:
            f = block.permute(0, 3, 1, 2).to(torch.float32).div_(255.0)
            if out is not None:
                out[: f.shape[0]].copy_(f)
                f = out[: f.shape[0]]
            block = f
        if out_labels is not None:
            out_labels[: labels.shape[0]].copy_(labels)
            labels = out_labels[: labels.shape[0]]
        return block, labels


def _synthetic_pool(pool: int, h: int, w: int, num_classes: int, seed: int):
    """Class-dependent synthetic images: a per-class mean colour pattern plus
    noise, so a model can actually fit them (loss-decrease tests)."""
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, num_classes, (pool,), generator=g)
    proto = torch.randint(32, 224, (num_classes, 4, 4, 3), generator=g, dtype=torch.int16)
    base = proto[labels]  # [pool,4,4,3]
    base = base.repeat_interleave(max(1, h // 4), 1).repeat_interleave(max(1, w // 4), 2)[:, :h, :w]
    if base.shape[1] < h or base.shape[2] < w:
        base = torch.nn.functional.pad(base.permute(0, 3, 1, 2).float(), (0, w - base.shape[2], 0, h - base.shape[1]),
                                       mode="replicate").permute(0, 2, 3, 1).to(torch.int16)
    noise = torch.randint(-24, 25, (pool, h, w, 3), generator=g, dtype=torch.int16)
    imgs = (base + noise).clamp_(0, 255).to(torch.uint8)
    return imgs, labels


class SyntheticCIFAR(_ArrayImageDataset):
    """CIFAR-shaped: 3x32x32, 50 000 train / 10 000 test, ``num_classes`` labels
    (100 like the reference; 10 for the CIFAR-10-shaped plumbing config)."""

    def __init__(self, train: bool = True, num_classes: int = 100, length: int | None = None,
                 pool: int | None = None, seed: int = 0):
        n = length if length is not None else (50_000 if train else 10_000)
        pool = min(n, pool if pool is not None else n)
        imgs, labels = _synthetic_pool(pool, 32, 32, num_classes, seed + (0 if train else 1))
        super().__init__(imgs, labels, n)
        self.num_classes = num_classes


class SyntheticImageNet(_ArrayImageDataset):
    """ImageNet-shaped: 3x224x224, 1000 classes.  A pool of ``pool`` distinct
    images is cycled to the nominal length (1 281 167 by default)."""

    def __init__(self, train: bool = True, num_classes: int = 1000, length: int | None = None,
                 pool: int = 1024, image_size: int = 224, seed: int = 0):
        n = length if length is not None else (1_281_167 if train else 50_000)
        pool = min(n, pool)
        imgs, labels = _synthetic_pool(pool, image_size, image_size, num_classes, seed + (0 if train else 1))
        super().__init__(imgs, labels, n)
        self.num_classes = num_classes


class CIFAR100Binary(_ArrayImageDataset):
    """Real CIFAR-100 from ``<root>/cifar-100-binary/{train,test}.bin``.

    Record = 1 coarse label byte, 1 fine label byte, 3072 pixel bytes (CHW).
    Fine labels are used, like torchvision's ``CIFAR100``.
    """

    def __init__(self, root: str = "dataset", train: bool = True):
        path = os.path.join(root, "cifar-100-binary", "train.bin" if train else "test.bin")
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found (no network here: place the CIFAR-100 binary files there, "
                f"or use --dataset synthetic-cifar)")
        raw = np.fromfile(path, dtype=np.uint8).reshape(-1, 3074)
        labels = torch.from_numpy(raw[:, 1].astype(np.int64))
        imgs = torch.from_numpy(raw[:, 2:].reshape(-1, 3, 32, 32)).permute(0, 2, 3, 1).contiguous()
        super().__init__(imgs, labels)
        self.num_classes = 100


# ---- the cifar-100-python layout without unpickling code ------------------------------------------------------------
# torchvision's CIFAR100 leaves ``<root>/cifar-100-python/{train,test,meta}``: Python-2 pickles of a dict
# {b"data": uint8 ndarray [N, 3072], b"fine_labels": [int], b"coarse_labels", b"filenames", b"batch_label"}.  A pickle
# can name ANY importable callable; this reader lets the stream name exactly three -- the ones a numpy array pickles
# as -- and resolves each to an inert stand-in defined here, so no function of numpy (or anything else) is called
# on behalf of the file.  Everything else in the stream must be plain containers, bytes, str and ints; any other
# global, a persistent id or an extension code raises.  The array is rebuilt from the raw buffer afterwards, with
# its dtype and shape checked.
import io
import pickle


class _NdStub:
    """What ``numpy.core.multiarray._reconstruct`` + BUILD leave behind: the pickled array's state tuple."""

    def __init__(self, *args):
        self.state = None

    def __setstate__(self, state):
        self.state = state


class _DtypeStub:
    def __init__(self, *args):
        self.args, self.state = args, None

    def __setstate__(self, state):
        self.state = state


def _nd_reconstruct(cls, shape, typecode):
    if cls is not _NdStub:
        raise pickle.UnpicklingError("cifar-100-python: unexpected array class")
    return _NdStub()


def _latin1_bytes(text, encoding="latin1"):
    """How Python 3 writes a ``bytes`` object at pickle protocol <= 2 (``_codecs.encode(str, 'latin1')``); the file the
    reference's download leaves was written by Python 2 and holds plain string opcodes instead."""
    if not isinstance(text, str) or str(encoding).lower().replace("-", "") != "latin1":
        raise pickle.UnpicklingError("cifar-100-python: unexpected text encoding request")
    return text.encode("latin1")


_ALLOWED_GLOBALS = {
    ("_codecs", "encode"): _latin1_bytes,
    ("numpy.core.multiarray", "_reconstruct"): _nd_reconstruct,
    ("numpy._core.multiarray", "_reconstruct"): _nd_reconstruct,
    ("numpy", "ndarray"): _NdStub,
    ("numpy", "dtype"): _DtypeStub,
}


class _InertUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return _ALLOWED_GLOBALS[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(f"cifar-100-python: the file names {module}.{name}; only plain containers and "
                                         f"numpy arrays are read") from None

    def persistent_load(self, pid):
        raise pickle.UnpicklingError("cifar-100-python: persistent ids are not read")


def _array_from_stub(stub: _NdStub, name: str) -> np.ndarray:
    st = stub.state
    if not (isinstance(st, tuple) and len(st) == 5):
        raise ValueError(f"cifar-100-python: {name}: unexpected array state")
    _, shape, dt, fortran, raw = st
    if not (isinstance(dt, _DtypeStub) and dt.args and dt.args[0] in ("u1", b"u1", "uint8", "|u1")):
        raise ValueError(f"cifar-100-python: {name}: expected a uint8 array")
    if isinstance(raw, str):  # (a Python-2 str decoded as latin-1)
        raw = raw.encode("latin1")
    if not (isinstance(shape, tuple) and all(isinstance(d, int) and d >= 0 for d in shape)) or fortran:
        raise ValueError(f"cifar-100-python: {name}: unexpected array layout")
    if int(np.prod(shape, dtype=np.int64)) != len(raw):
        raise ValueError(f"cifar-100-python: {name}: {len(raw)} bytes for shape {shape}")
    return np.frombuffer(raw, dtype=np.uint8).reshape(shape)


def _load_cifar_python(path: str):
    """-> (uint8 [N, 3072] array, list of fine labels) from a ``cifar-100-python`` batch file."""
    with open(path, "rb") as fh:
        obj = _InertUnpickler(io.BytesIO(fh.read()), encoding="bytes").load()
    if not isinstance(obj, dict):
        raise ValueError("cifar-100-python: the file does not hold a dict")
    obj = {(k.decode() if isinstance(k, bytes) else k): v for k, v in obj.items()}
    data, labels = obj.get("data"), obj.get("fine_labels")
    if not isinstance(data, _NdStub) or not isinstance(labels, list) or not all(isinstance(v, int) for v in labels):
        raise ValueError("cifar-100-python: expected keys data (uint8 array) and fine_labels (list of ints)")
    arr = _array_from_stub(data, "data")
    if arr.ndim != 2 or arr.shape[1] != 3072 or arr.shape[0] != len(labels):
        raise ValueError(f"cifar-100-python: data {arr.shape} does not match {len(labels)} labels of 3x32x32 images")
    return arr, labels


class CIFAR100Python(_ArrayImageDataset):
    """Real CIFAR-100 from ``<root>/cifar-100-python/{train,test}``, the layout ``torchvision.datasets.CIFAR100(
    root, download=True)`` of the reference (``/root/reference/main.py:42-51``) leaves on disk.  Fine labels."""

    def __init__(self, root: str = "dataset", train: bool = True):
        path = os.path.join(root, "cifar-100-python", "train" if train else "test")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found")
        arr, labels = _load_cifar_python(path)
        imgs = torch.from_numpy(arr.reshape(-1, 3, 32, 32).copy()).permute(0, 2, 3, 1).contiguous()
        super().__init__(imgs, torch.tensor(labels, dtype=torch.int64))
        self.num_classes = 100


def build_dataset(name: str, train: bool = True, root: str = "dataset", num_classes: int | None = None,
                  length: int | None = None, seed: int = 0):
    name = name.lower()
    if name in ("cifar100", "cifar-100"):
        path = os.path.join(root, "cifar-100-binary", "train.bin" if train else "test.bin")
        py_path = os.path.join(root, "cifar-100-python", "train" if train else "test")
        kind = 2 if os.path.exists(path) else (1 if os.path.exists(py_path) else 0)  # binary preferred
        found = kind > 0
        # every rank must train on the same source: agree on "every rank has the files" (one MIN all-reduce), so a
        # node without them turns the whole job to the synthetic set instead of silently feeding different data
        import torch.distributed as tdist

        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
            flag = torch.tensor([kind], dtype=torch.int32, device=dev)
            tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)  # (the two layouts hold the same images in the same order)
            everywhere = int(flag.item()) > 0
            if found and not everywhere:
                sys.stderr.write(f"[pdt] CIFAR-100 files exist under {root} here but not on every rank: all ranks use "
                                 f"synthetic data\n")
            found = everywhere
        if found:
            return CIFAR100Binary(root, train) if kind == 2 else CIFAR100Python(root, train)
        if not tdist.is_initialized() or tdist.get_rank() == 0:
            sys.stderr.write(f"[pdt] NOTICE: neither {path} nor {py_path} found (no network to download CIFAR-100): "
                             f"using SYNTHETIC CIFAR-100-shaped data; results are not CIFAR-100 results\n")
            sys.stderr.flush()
        # same shapes, dtypes and length as the reference's dataset
        return SyntheticCIFAR(train, num_classes or 100, length, seed=seed)
    if name in ("synthetic-cifar", "synthetic_cifar"):
        return SyntheticCIFAR(train, num_classes or 100, length, seed=seed)
    if name in ("synthetic-cifar10", "synthetic_cifar10"):
        return SyntheticCIFAR(train, num_classes or 10, length, seed=seed)
    if name in ("synthetic-imagenet", "synthetic_imagenet"):
        return SyntheticImageNet(train, num_classes or 1000, length, seed=seed)
    raise ValueError(f"unknown dataset {name!r}")
