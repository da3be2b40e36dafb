"""Per-GPU batch sizing for 288 GB of HBM3E (BASELINE.json config 3).

One ``gpurun`` round trip costs minutes and an out-of-memory kill can take the box down, so the batch is
chosen from a model instead of by trial and error: two short probe steps give the activation bytes per image
(the slope of peak memory in the batch) and the fixed cost (weights, optimizer state, buckets, workspaces),
forward hooks give the largest single activation per image, and the batch is the largest that fits ``fraction``
of the device memory.

The buffer descriptors the convolution / pooling kernels address an operand with span 2 GiB per LAUNCH, not per
tensor: the bindings cut a larger batch into chunks of images (csrc/binding/ops.cpp ``chunk_images``), so the
descriptor range only requires that ONE image of the widest activation fits (``by_tensor_range`` is reported as
the images per launch of that activation: below it a layer runs as one launch, above it as several).
"""
from __future__ import annotations

import torch

_TENSOR_LIMIT = (1 << 31) - 4096  # bytes of one operand a single launch may span (32-bit buffer offsets)


def _probe_step(engine, batch: int, image_shape, num_classes: int):
    dev = engine.device
    x = torch.rand(batch, *image_shape, device=dev)
    y = torch.randint(0, num_classes, (batch,), device=dev)
    torch.cuda.synchronize(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    engine._eager_step(x, y)
    torch.cuda.synchronize(dev)
    return torch.cuda.max_memory_allocated(dev)


def suggest_batch(engine, image_shape, num_classes: int, probe: int = 16, fraction: float = 0.85,
                  multiple: int = 64, cap: int | None = None) -> dict:
    """Largest per-GPU batch (a multiple of ``multiple``) that the memory model allows for ``engine``'s network.
    Returns the pick and the model's terms; leaves the engine trained by three probe steps."""
    if engine.device.type != "cuda":
        raise RuntimeError("batch sizing models GPU memory; pick the batch by hand on CPU")
    biggest = {"numel": 0}
    hooks = []

    def hook(_m, _inp, out):
        if torch.is_tensor(out) and out.dim() >= 2:
            biggest["numel"] = max(biggest["numel"], out.numel() // out.shape[0])

    for m in engine.net.modules():
        if not list(m.children()):
            hooks.append(m.register_forward_hook(hook))
    try:
        _probe_step(engine, probe, image_shape, num_classes)  # warm-up: lazy state, reducer rebuild, workspaces
        p1 = _probe_step(engine, probe, image_shape, num_classes)
        p2 = _probe_step(engine, 2 * probe, image_shape, num_classes)
    finally:
        for h in hooks:
            h.remove()
    per_image = max(1.0, (p2 - p1) / probe)
    fixed = max(0.0, p1 - per_image * probe)
    total = torch.cuda.get_device_properties(engine.device).total_memory
    by_memory = int((total * fraction - fixed) / per_image)
    elem = 2 if engine.dtype in (torch.bfloat16, torch.float16) else 4
    # the packed stem image is the widest per-image tensor some networks make: (H+6) x (W+8) x 4 elements
    widest = max(biggest["numel"], (image_shape[-2] + 6) * (image_shape[-1] + 8) * 4 if len(image_shape) == 3 else 0)
    by_tensor = _TENSOR_LIMIT // max(1, widest * elem)
    if by_tensor < 1:
        raise RuntimeError(f"one image of the widest activation ({widest * elem} bytes) exceeds a buffer descriptor")
    pick = by_memory
    if cap is not None:
        pick = min(pick, cap)
    pick = max(multiple, pick // multiple * multiple) if pick >= multiple else max(1, pick)
    return {"batch": int(pick), "by_memory": by_memory, "by_tensor_range": int(by_tensor),
            "bytes_per_image": per_image, "fixed_bytes": fixed, "device_bytes": int(total),
            "widest_activation_elems_per_image": int(widest)}
