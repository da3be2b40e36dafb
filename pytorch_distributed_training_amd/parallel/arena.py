"""Flat parameter / buffer arenas and the bf16 working copy.

MI355X design: parameters live in ONE
contiguous fp32 arena per module, so that

* the initial state broadcast is one ``ncclBroadcast`` instead of 161 + 159
  coalesced tensors,
* the fused optimizer walks large contiguous ranges (one launch),
* the bf16 working copy that the MFMA kernels read is a second flat arena that
  the optimizer kernel refreshes in the same pass (fp32 master weights, bf16
  compute).

Conv weights are stored physically ``[Cout][kh][kw][Cin]`` (torch
``channels_last``): the K-contiguous operand layout of the implicit-GEMM
kernels.  Logical shapes are unchanged, so ``state_dict`` round-trips with
stock models (``load_state_dict`` copies INTO the arena; Parameter identity is
preserved, so an optimizer built before flattening stays valid —
``/root/reference/main.py:80-83`` builds Adam before ``.cuda()`` and DDP).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as tnn

_ALIGN = 128  # elements; keeps every slot 256 B (bf16) / 512 B (fp32) aligned for 16 B/lane access


def _round_up(n: int, a: int = _ALIGN) -> int:
    return (n + a - 1) // a * a


@dataclass
class Slot:
    name: str
    offset: int
    numel: int
    shape: tuple
    channels_last: bool


def _view(flat: torch.Tensor, slot: Slot) -> torch.Tensor:
    seg = flat[slot.offset: slot.offset + slot.numel]
    if slot.channels_last:
        co, ci, kh, kw = slot.shape
        return seg.view(co, kh, kw, ci).permute(0, 3, 1, 2)
    return seg.view(slot.shape)


class ParamArena:
    """Flat storage for the parameters of a module (one arena per dtype)."""

    def __init__(self, module: tnn.Module, channels_last_weights: bool | None = None):
        params = [(n, p) for n, p in module.named_parameters()]
        if not params:
            raise ValueError("module has no parameters")
        self.device = params[0][1].device
        if channels_last_weights is None:
            channels_last_weights = self.device.type == "cuda"
        self.by_dtype: dict[torch.dtype, torch.Tensor] = {}
        self.slots: dict[str, tuple[torch.dtype, Slot]] = {}
        self.params: list[tnn.Parameter] = []
        self.names: list[str] = []
        groups: dict[torch.dtype, list[tuple[str, tnn.Parameter]]] = {}
        seen: set[int] = set()
        for n, p in params:
            if id(p) in seen:
                continue
            seen.add(id(p))
            if p.device != self.device:
                raise ValueError(f"parameter {n} on {p.device}, expected {self.device} (one device per process)")
            groups.setdefault(p.dtype, []).append((n, p))
        for dtype, plist in groups.items():
            total, slots = 0, []
            for n, p in plist:
                cl = channels_last_weights and p.dim() == 4
                slots.append(Slot(n, total, p.numel(), tuple(p.shape), cl))
                total += _round_up(p.numel())
            flat = torch.zeros(total, dtype=dtype, device=self.device)
            with torch.no_grad():
                for (n, p), s in zip(plist, slots):
                    v = _view(flat, s)
                    v.copy_(p.data)
                    p.data = v
                    p._arena_slot = (dtype, s.offset, s.numel)
                    self.slots[n] = (dtype, s)
                    self.params.append(p)
                    self.names.append(n)
            self.by_dtype[dtype] = flat
        self.lowp: dict[torch.dtype, torch.Tensor] = {}
        self.lowp_dtype: torch.dtype | None = None

    # -- bf16 working copy ---------------------------------------------------
    def attach_lowp(self, dtype: torch.dtype = torch.bfloat16) -> None:
        """Create the flat low-precision mirror and hang per-parameter views on
        ``param._lp``.  Only fp32 parameters are mirrored."""
        self.lowp_dtype = dtype
        flat = self.by_dtype.get(torch.float32)
        if flat is None:
            return
        lp = flat.to(dtype)
        self.lowp[torch.float32] = lp
        for n, p in zip(self.names, self.params):
            dt, s = self.slots[n]
            if dt == torch.float32:
                p._lp = _view(lp, s)
                p._lp._pdt_param = True  # long-lived: lets the op layer cache derived layouts per step

    @torch.no_grad()
    def refresh_lowp(self) -> None:
        """Re-cast master -> working copy (the fused optimizer does this in-kernel;
        this is for loads / manual weight edits)."""
        for dt, lp in self.lowp.items():
            lp.copy_(self.by_dtype[dt])
        if self.lowp and self.device.type == "cuda":
            from ..ops import native

            native.bump_weights_epoch()

    def flat_tensors(self) -> list[torch.Tensor]:
        return list(self.by_dtype.values())

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.by_dtype.values())


class BufferArena:
    """Flat storage for module buffers (BN running stats; ``num_batches_tracked``)."""

    def __init__(self, module: tnn.Module):
        self.by_dtype: dict[torch.dtype, torch.Tensor] = {}
        groups: dict[torch.dtype, list] = {}
        seen: set[int] = set()
        for mod in module.modules():
            for bname, b in list(mod._buffers.items()):
                if b is None or id(b) in seen:
                    continue
                seen.add(id(b))
                groups.setdefault(b.dtype, []).append((mod, bname, b))
        for dtype, items in groups.items():
            total, offs = 0, []
            for _, _, b in items:
                offs.append(total)
                total += _round_up(max(b.numel(), 1), 16)
            flat = torch.zeros(total, dtype=dtype, device=items[0][2].device)
            for (mod, bname, b), off in zip(items, offs):
                v = flat[off: off + b.numel()].view(b.shape)
                v.copy_(b)
                # keep the Tensor object (others may hold it: convert_sync_batchnorm shares buffers)
                b.set_(v.untyped_storage(), v.storage_offset(), v.shape, v.stride())
            self.by_dtype[dtype] = flat

    def flat_tensors(self) -> list[torch.Tensor]:
        return list(self.by_dtype.values())
