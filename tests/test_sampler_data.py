"""T0: sampler index parity with the stock sampler, datasets, loader."""
import math

import pytest
import torch
from torch.utils.data.distributed import DistributedSampler as TorchSampler

from pytorch_distributed_training_amd.data import (DataLoader, DistributedSampler, SyntheticCIFAR,
                                                   SyntheticImageNet, build_dataset)


class _DS:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


@pytest.mark.parametrize("n", [1, 7, 100, 1000, 50_000])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("shuffle", [True, False])
@pytest.mark.parametrize("drop_last", [True, False])
def test_sampler_matches_torch(n, world, shuffle, drop_last):
    if drop_last and n < world:
        pytest.skip("torch yields negative length here")
    ds = _DS(n)
    for rank in {0, world - 1, world // 2}:
        for epoch, seed in ((0, 0), (3, 0), (1, 17)):
            ours = DistributedSampler(ds, world, rank, shuffle=shuffle, seed=seed, drop_last=drop_last)
            ref = TorchSampler(ds, world, rank, shuffle=shuffle, seed=seed, drop_last=drop_last)
            ours.set_epoch(epoch)
            ref.set_epoch(epoch)
            assert len(ours) == len(ref)
            assert list(ours) == list(ref)


def test_reference_iteration_counts():
    """50 000 samples, B=128, drop_last=False everywhere."""
    expect = {1: (391, 80), 2: (196, 40), 4: (98, 84), 8: (49, 106)}
    ds = _DS(50_000)
    for w, (batches, last) in expect.items():
        s = DistributedSampler(ds, w, 0)
        assert len(s) == math.ceil(50_000 / w)
        dl = DataLoader(ds, batch_size=128, sampler=s)
        assert len(dl) == batches
        assert len(s) - (batches - 1) * 128 == last


def test_sampler_shards_cover_dataset():
    ds = _DS(103)
    seen = []
    for r in range(4):
        s = DistributedSampler(ds, 4, r, shuffle=True, seed=5)
        s.set_epoch(2)
        seen += list(s)
    assert set(seen) == set(range(103)) and len(seen) == 104


def test_sampler_resume_state():
    ds = _DS(50)
    s = DistributedSampler(ds, 2, 1)
    s.set_epoch(4)
    full = list(s)
    t = DistributedSampler(ds, 2, 1)
    t.load_state_dict(s.state_dict())
    assert list(t.iter_from(7)) == full[7:]


def test_synthetic_cifar_shapes():
    ds = SyntheticCIFAR(train=True, num_classes=100, length=64)
    img, label = ds[3]
    assert img.shape == (3, 32, 32) and img.dtype == torch.float32
    assert 0.0 <= float(img.min()) and float(img.max()) <= 1.0
    assert isinstance(label, int) and 0 <= label < 100
    assert len(SyntheticCIFAR(train=True, pool=16)) == 50_000
    assert len(SyntheticCIFAR(train=False, pool=16)) == 10_000


def test_synthetic_imagenet_shapes():
    ds = SyntheticImageNet(pool=4, length=10)
    img, label = ds[9]
    assert img.shape == (3, 224, 224) and 0 <= label < 1000
    x, y = ds.get_batch([0, 5, 9])
    assert x.shape == (3, 3, 224, 224) and y.shape == (3,)
    u, _ = ds.get_batch([1, 2], raw_u8=True)
    assert u.dtype == torch.uint8 and u.shape == (2, 224, 224, 3)


def test_get_batch_matches_getitem():
    ds = SyntheticCIFAR(length=40, num_classes=10)
    x, y = ds.get_batch([3, 7, 39])
    for k, i in enumerate([3, 7, 39]):
        img, lab = ds[i]
        assert torch.equal(x[k], img) and int(y[k]) == lab


def test_cifar100_falls_back_to_synthetic(tmp_path):
    ds = build_dataset("cifar100", root=str(tmp_path), length=32)
    assert len(ds) == 32 and ds.sample_shape == (3, 32, 32)


def test_cifar100_binary_reader(tmp_path):
    import numpy as np

    from pytorch_distributed_training_amd.data import CIFAR100Binary

    d = tmp_path / "cifar-100-binary"
    d.mkdir()
    rng = np.random.default_rng(0)
    rec = rng.integers(0, 256, size=(5, 3074), dtype=np.uint8)
    rec[:, 1] = [0, 99, 5, 7, 42]
    rec.tofile(d / "train.bin")
    ds = CIFAR100Binary(str(tmp_path), train=True)
    assert len(ds) == 5
    img, lab = ds[1]
    assert lab == 99
    want = torch.from_numpy(rec[1, 2:].reshape(3, 32, 32).astype("float32") / 255.0)
    assert torch.allclose(img, want)


@pytest.mark.parametrize("prefetch", [0, 2])
def test_loader_order_and_short_last_batch(prefetch):
    ds = SyntheticCIFAR(length=70, num_classes=10)
    s = DistributedSampler(ds, 2, 0, shuffle=True)
    s.set_epoch(1)
    dl = DataLoader(ds, batch_size=16, sampler=s, prefetch=prefetch)
    idx = list(s)
    batches = list(dl)
    assert [b[0].shape[0] for b in batches] == [16, 16, 3]
    got = torch.cat([b[1] for b in batches])
    want = torch.tensor([ds[i][1] for i in idx])
    assert torch.equal(got, want)
    first_img = ds[idx[0]][0]
    assert torch.equal(batches[0][0][0], first_img)


def test_loader_generic_dataset_collate():
    class Plain:
        def __len__(self):
            return 5

        def __getitem__(self, i):
            return torch.full((3, 2, 2), float(i)), i

    out = list(DataLoader(Plain(), batch_size=2))
    assert [o[0].shape[0] for o in out] == [2, 2, 1]
    assert out[1][1].tolist() == [2, 3]


def test_loader_rejects_sampler_and_shuffle():
    ds = SyntheticCIFAR(length=8)
    with pytest.raises(ValueError):
        DataLoader(ds, batch_size=2, shuffle=True, sampler=DistributedSampler(ds, 1, 0))


def test_loader_raw_uint8_batches_match_the_totensor_path():
    """Batches bound for a GPU travel as uint8 NHWC; Engine.prepare_input turns them into what the fp32 path gives."""
    import torch

    from pytorch_distributed_training_amd.data import DataLoader
    from pytorch_distributed_training_amd.data.datasets import SyntheticCIFAR
    from pytorch_distributed_training_amd.engine import Engine

    ds = SyntheticCIFAR(train=True, num_classes=10, length=50)
    a = list(DataLoader(ds, batch_size=16))
    b = list(DataLoader(ds, batch_size=16, raw_u8=True))
    assert DataLoader(ds, batch_size=16).raw_u8 is False  # host-side consumers keep ToTensor's fp32 NCHW
    assert len(a) == len(b) == 4 and b[0][0].dtype == torch.uint8 and tuple(b[0][0].shape) == (16, 32, 32, 3)
    assert tuple(b[-1][0].shape) == (2, 32, 32, 3)  # short last batch
    eng = Engine.__new__(Engine)
    eng.device, eng.dtype = torch.device("cpu"), torch.float32
    for (xa, ya), (xb, yb) in zip(a, b):
        assert torch.equal(ya, yb)
        assert torch.allclose(Engine.prepare_input(eng, xb), xa, atol=1e-7)


def _py2_cifar_pickle(data, fine):
    """A pickle as Python 2's cPickle (protocol 2) writes the CIFAR-100 batch dict: string opcodes for keys and the
    raw buffer, ``numpy.core.multiarray._reconstruct`` + BUILD for the array (what torchvision's download leaves)."""
    import struct

    def sstr(b):  # SHORT_BINSTRING
        return b"U" + bytes([len(b)]) + b
    n, w = data.shape
    arr = (b"cnumpy.core.multiarray\n_reconstruct\n" + b"cnumpy\nndarray\n" + b"K\x00\x85" + sstr(b"b") + b"\x87R"
           + b"(K\x01" + b"(" + b"K" + bytes([n]) + b"M" + struct.pack("<H", w) + b"t"
           + b"cnumpy\ndtype\n" + sstr(b"u1") + b"K\x00K\x01\x87R" + b"(K\x03" + sstr(b"|") + b"NNN" + b"J\xff\xff\xff\xff" * 2
           + b"K\x00tb" + b"\x89" + b"T" + struct.pack("<i", data.size) + data.tobytes() + b"tb")
    labels = b"]" + b"(" + b"".join(b"K" + bytes([v]) for v in fine) + b"e"
    return (b"\x80\x02}(" + sstr(b"data") + arr + sstr(b"fine_labels") + labels + sstr(b"coarse_labels") + labels
            + sstr(b"batch_label") + sstr(b"training batch 1 of 1") + b"u.")


def test_cifar100_python_layout_is_read_without_running_pickled_code(tmp_path):
    """The tree the reference's ``CIFAR100(root='dataset', download=True)`` leaves (``/root/reference/main.py:42-51``):
    read through an unpickler whose only resolvable globals are inert stand-ins for numpy's array pickling."""
    import os
    import pickle

    import numpy as np

    from pytorch_distributed_training_amd.data import build_dataset
    from pytorch_distributed_training_amd.data.datasets import CIFAR100Python

    rng = np.random.default_rng(0)
    data = rng.integers(0, 256, (12, 3072), dtype=np.uint8)
    fine = [int(v) for v in rng.integers(0, 100, 12)]
    d = tmp_path / "cifar-100-python"
    d.mkdir()
    for name in ("train", "test"):
        (d / name).write_bytes(_py2_cifar_pickle(data, fine))
    pickle.loads(_py2_cifar_pickle(data, fine), encoding="bytes")  # the hand-built stream is a valid numpy pickle
    ds = build_dataset("cifar100", train=True, root=str(tmp_path))
    assert isinstance(ds, CIFAR100Python) and len(ds) == 12 and ds.num_classes == 100
    img, label = ds[5]
    assert label == fine[5] and img.shape == (3, 32, 32)
    assert np.array_equal((img * 255).round().numpy().astype(np.uint8), data[5].reshape(3, 32, 32))
    # a Python-3 protocol-4 dump of the same dict reads the same
    (d / "train").write_bytes(pickle.dumps({b"data": data, b"fine_labels": fine}, protocol=4))
    assert torch.equal(CIFAR100Python(str(tmp_path)).images, ds.images)

    # anything else a pickle could name is refused before it is called
    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > " + str(tmp_path / "pwned"),))

    (d / "train").write_bytes(pickle.dumps({b"data": Evil(), b"fine_labels": fine}, protocol=2))
    with pytest.raises(pickle.UnpicklingError, match="only plain containers"):
        CIFAR100Python(str(tmp_path))
    assert not (tmp_path / "pwned").exists()
    # a float64 array where the images should be
    (d / "train").write_bytes(pickle.dumps({b"data": data.astype(np.float64), b"fine_labels": fine}, protocol=2))
    with pytest.raises(ValueError, match="uint8"):
        CIFAR100Python(str(tmp_path))
