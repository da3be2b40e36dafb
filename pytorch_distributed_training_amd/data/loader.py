"""Batch loader: sampler-driven order, pinned ring, async H2D prefetch.

Reference: ``DataLoader(trainset, batch_size=batch_size, shuffle=False,
pin_memory=True, sampler=sampler)`` with ``num_workers=0`` and
``drop_last=False`` (short last batch), followed by *blocking*
``imgs.cuda(); labels.cuda()`` each step (``/root/reference/main.py:54-58,
98-99``).

MI355X design: a background thread assembles batch ``t+1`` into a ring of
pinned host buffers while step ``t`` runs; when ``device`` is given the H2D
copy of batch ``t+1`` is issued on a dedicated HIP copy stream and fenced to
the compute stream with an event, so the copy overlaps compute instead of
stalling the step.  Batches bound for a GPU travel as uint8 NHWC (a quarter of
the bytes; the device-side cast kernel does /255, bf16 and the layout).  Order,
batch boundaries and the short last batch are identical to the reference loader.
"""
from __future__ import annotations

import queue
import threading
from typing import Iterable, Iterator

import torch


def default_collate(samples):
    imgs = torch.stack([s[0] for s in samples], 0)
    labels = torch.as_tensor([s[1] for s in samples], dtype=torch.int64)
    return imgs, labels


class _SequentialSampler:
    def __init__(self, n: int):
        self.n = n

    def __iter__(self):
        return iter(range(self.n))

    def __len__(self):
        return self.n


class _RandomSampler(_SequentialSampler):
    def __iter__(self):
        return iter(torch.randperm(self.n).tolist())


class DataLoader:
    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False, sampler: Iterable[int] | None = None,
                 drop_last: bool = False, pin_memory: bool = False, prefetch: int = 2,
                 device: torch.device | str | None = None, collate_fn=None, start_sample: int = 0,
                 raw_u8: bool | None = None):
        if sampler is not None and shuffle:
            raise ValueError("sampler option is mutually exclusive with shuffle")
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.drop_last = drop_last
        self.sampler = sampler if sampler is not None else (
            _RandomSampler(len(dataset)) if shuffle else _SequentialSampler(len(dataset)))
        self.pin_memory = bool(pin_memory) and torch.cuda.is_available()
        self.prefetch = max(0, int(prefetch))
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.type != "cuda":
            self.device = None
        self.collate_fn = collate_fn or default_collate
        self.start_sample = int(start_sample)  # mid-epoch resume: skip this many samples once
        # uint8 NHWC batches (the decoded-image layout) when they go to a GPU: a quarter of the host gather, the pinned
        # ring and the H2D bytes; /255 + bf16 + layout happen in the engine's one cast kernel on the device
        # (Engine.prepare_input).  Off for host-side consumers, which expect ToTensor's fp32 NCHW.
        can_raw = hasattr(dataset, "get_batch") and hasattr(dataset, "images") and self.collate_fn is default_collate
        self.raw_u8 = (self.device is not None and can_raw) if raw_u8 is None else (bool(raw_u8) and can_raw)
        self._ring: list[tuple[torch.Tensor, torch.Tensor]] = []
        self._slot_events: dict[int, "torch.cuda.Event"] = {}  # H2D completion per pinned slot
        self._copy_stream = None

    def __len__(self) -> int:
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # ---- host side -------------------------------------------------------
    def _index_batches(self) -> Iterator[list[int]]:
        if self.start_sample and hasattr(self.sampler, "iter_from"):
            it = self.sampler.iter_from(self.start_sample)
            self.start_sample = 0
        else:
            it = iter(self.sampler)
        batch: list[int] = []
        for i in it:
            batch.append(i)
            if len(batch) == self.batch_size:
                yield batch
                batch = []
        if batch and not self.drop_last:
            yield batch

    def _ring_slot(self, k: int, sample_shape, img_dtype):
        while len(self._ring) <= k:
            self._ring.append((
                torch.empty((self.batch_size, *sample_shape), dtype=img_dtype).pin_memory(),
                torch.empty((self.batch_size,), dtype=torch.int64).pin_memory(),
            ))
        return self._ring[k]

    def _host_batch(self, indices: list[int], slot: int):
        ev = self._slot_events.pop(slot, None)
        if ev is not None:
            ev.synchronize()  # the previous H2D out of this pinned slot must have drained
        return (*self._host_batch_impl(indices, slot), slot)

    def _host_batch_impl(self, indices: list[int], slot: int):
        fast = hasattr(self.dataset, "get_batch") and self.collate_fn is default_collate
        if fast:
            if self.raw_u8:
                if self.pin_memory:
                    c, h, w = self.dataset.sample_shape
                    out, out_l = self._ring_slot(slot, (h, w, c), torch.uint8)
                    return self.dataset.get_batch(indices, out=out, out_labels=out_l, raw_u8=True)
                return self.dataset.get_batch(indices, raw_u8=True)
            if self.pin_memory:
                out, out_l = self._ring_slot(slot, self.dataset.sample_shape, torch.float32)
                return self.dataset.get_batch(indices, out=out, out_labels=out_l)
            return self.dataset.get_batch(indices)
        imgs, labels = self.collate_fn([self.dataset[i] for i in indices])
        if self.pin_memory:
            out, out_l = self._ring_slot(slot, tuple(imgs.shape[1:]), imgs.dtype)
            out[: imgs.shape[0]].copy_(imgs)
            out_l[: labels.shape[0]].copy_(labels)
            imgs, labels = out[: imgs.shape[0]], out_l[: labels.shape[0]]
        return imgs, labels

    def _host_iter(self):
        nslots = self.prefetch + 2  # one being filled, `prefetch` queued, one in use by the consumer
        if self.prefetch == 0:
            for k, idx in enumerate(self._index_batches()):
                yield self._host_batch(idx, k % nslots)
            return
        q: queue.Queue = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        _END = object()

        def work():
            try:
                for k, idx in enumerate(self._index_batches()):
                    if stop.is_set():
                        return
                    item = self._host_batch(idx, k % nslots)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(_END)
            except BaseException as e:  # surface loader errors in the consumer
                q.put(e)

        t = threading.Thread(target=work, name="pdt-loader", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()

    # ---- device side -----------------------------------------------------
    def _device_iter(self):
        dev = self.device
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        cs = self._copy_stream

        def upload(item):
            imgs, labels, slot = item
            with torch.cuda.stream(cs):
                d_imgs = imgs.to(dev, non_blocking=True)
                d_labels = labels.to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cs)
            if self.pin_memory:
                self._slot_events[slot] = ev
            return d_imgs, d_labels, ev

        it = self._host_iter()
        try:
            nxt = upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            d_imgs, d_labels, ev = nxt
            try:
                nxt = upload(next(it))  # batch t+1 starts moving before batch t is consumed
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            d_imgs.record_stream(cur)
            d_labels.record_stream(cur)
            yield d_imgs, d_labels

    def __iter__(self):
        if self.device is not None:
            return self._device_iter()
        return ((imgs, labels) for imgs, labels, _slot in self._host_iter())


class SyntheticDeviceLoader:
    """On-device synthetic batches of a fixed shape (benchmarks: no host data path).

    ``data: synthetic`` in bench.py: a small ring of pre-generated device
    batches is cycled, so every step still reads a distinct input tensor from
    HBM, but no H2D traffic or host work sits in the timed region.
    """

    def __init__(self, batch_size: int, sample_shape=(3, 224, 224), num_classes: int = 1000, steps: int | None = None,
                 device="cuda", dtype=torch.float32, ring: int = 4, seed: int = 0, channels_last: bool = False):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.steps = steps
        self.batches = []
        for _ in range(ring):
            x = torch.rand((batch_size, *sample_shape), generator=g, dtype=torch.float32)
            y = torch.randint(0, num_classes, (batch_size,), generator=g)
            x = x.to(device=device, dtype=dtype)
            if channels_last:
                x = x.contiguous(memory_format=torch.channels_last)
            self.batches.append((x, y.to(device)))

    def __len__(self):
        return self.steps if self.steps is not None else 1 << 62

    def __iter__(self):
        k = 0
        while self.steps is None or k < self.steps:
            yield self.batches[k % len(self.batches)]
            k += 1
