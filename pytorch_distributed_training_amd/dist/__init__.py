"""Process-group bootstrap and the few explicit collectives of the reference.

Reference behaviour (``/root/reference/main.py:34-37``)::

    dist.init_process_group(backend='nccl', init_method='env://')
    torch.cuda.set_device(args.local_rank)
    global_rank = dist.get_rank(); world_size = dist.get_world_size()

and ``reduce_loss`` (``main.py:16-20``).  Here the backend is selectable:
``rccl`` (alias ``nccl``; RCCL over xGMI, one process per GPU) or ``gloo``
(CPU plumbing, BASELINE.json config 1).  The rendezvous is ``env://``: rank 0
hosts a TCP store on ``MASTER_ADDR:MASTER_PORT``.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass

import torch
import torch.distributed as tdist

_BACKEND_ALIASES = {"rccl": "nccl", "nccl": "nccl", "gloo": "gloo"}


@dataclass
class DistEnv:
    rank: int
    world_size: int
    local_rank: int
    backend: str  # "rccl" or "gloo" (our naming)
    device: torch.device

    @property
    def is_main(self) -> bool:
        return self.rank == 0


_ENV: DistEnv | None = None


def resolve_local_rank(cli_value: int | None = None) -> int:
    """``--local_rank`` / ``--local-rank`` / ``LOCAL_RANK``."""
    if cli_value is not None:
        return int(cli_value)
    return int(os.environ.get("LOCAL_RANK", "0"))


def default_backend() -> str:
    return "rccl" if torch.cuda.is_available() else "gloo"


def init_process_group(
    backend: str | None = None,
    init_method: str = "env://",
    local_rank: int | None = None,
    timeout_s: float | None = None,
    device: str | None = None,
) -> DistEnv:
    """Initialise the default group; process group first, then device selection
    (the order of ``main.py:34-35``)."""
    global _ENV
    backend = (backend or default_backend()).lower()
    if backend not in _BACKEND_ALIASES:
        raise ValueError(f"unknown backend {backend!r}; choose rccl (alias nccl) or gloo")
    torch_backend = _BACKEND_ALIASES[backend]
    if torch_backend == "nccl" and not torch.cuda.is_available():
        raise RuntimeError("backend rccl needs a visible MI355X; use --backend gloo for CPU plumbing")
    local_rank = resolve_local_rank(local_rank)

    if init_method == "env://":
        # Single-process convenience: fill the env contract the launcher would
        # have provided so `python main.py` alone still works.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

    kwargs = {}
    if timeout_s is not None:
        kwargs["timeout"] = datetime.timedelta(seconds=timeout_s)
    if torch_backend == "nccl":
        # bind the communicator to this rank's GPU up front (avoids the lazy
        # "guess device from rank" path in ProcessGroupNCCL).
        kwargs["device_id"] = torch.device("cuda", local_rank)
    if not tdist.is_initialized():
        tdist.init_process_group(backend=torch_backend, init_method=init_method, **kwargs)

    # ``device="cuda"`` (or PDT_DEVICE=cuda) with the gloo backend keeps the tensors on the GPU and moves the
    # collectives through the host: a way to rehearse W > 1 on a box with ONE GPU (RCCL refuses two ranks on
    # one device).  Diagnostic only: see bench/micro/engine_w2_worker.py for what was observed.
    want = (device or os.environ.get("PDT_DEVICE", "auto")).lower()
    if want not in ("auto", "cuda", "cpu"):
        raise ValueError(f"unknown device {want!r}; choose auto, cuda or cpu")
    if torch.cuda.is_available() and (torch_backend == "nccl" or want == "cuda"):
        # (ranks may share a device only in the gloo rehearsal mode)
        local_dev = local_rank if torch_backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_dev)
        device = torch.device("cuda", local_dev)
    else:
        if want == "cuda":
            raise RuntimeError("device cuda requested but no GPU is visible")
        device = torch.device("cpu")
    _ENV = DistEnv(
        rank=tdist.get_rank(),
        world_size=tdist.get_world_size(),
        local_rank=local_rank,
        backend="rccl" if torch_backend == "nccl" else "gloo",
        device=device,
    )
    return _ENV


def env() -> DistEnv:
    if _ENV is None:
        if tdist.is_initialized():
            dev = torch.device("cuda", torch.cuda.current_device()) if (
                torch.cuda.is_available() and tdist.get_backend() == "nccl") else torch.device("cpu")
            return DistEnv(tdist.get_rank(), tdist.get_world_size(), resolve_local_rank(),
                           "rccl" if tdist.get_backend() == "nccl" else "gloo", dev)
        return DistEnv(0, 1, 0, "none", torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu"))
    return _ENV


def is_initialized() -> bool:
    return tdist.is_available() and tdist.is_initialized()


def get_rank() -> int:
    return tdist.get_rank() if is_initialized() else 0


def get_world_size() -> int:
    return tdist.get_world_size() if is_initialized() else 1


def barrier() -> None:
    if is_initialized() and get_world_size() > 1:
        if tdist.get_backend() == "nccl":
            tdist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            tdist.barrier()


def destroy_process_group() -> None:
    global _ENV
    if is_initialized():
        tdist.destroy_process_group()
    _ENV = None


def shutdown(engine=None, timeout_s: float = 60.0) -> None:
    """Ordered end of a (multi-rank) job, every step bounded by one watchdog timer:

    1. barrier: every rank is past its last collective;
    2. ``engine.close()``: reducer hooks off, device synchronised, the native communicator closed
       (watchdog thread joined -> communicator stream drained -> ``ncclCommDestroy``);
    3. any other native communicator of this process closed the same way, then the IPC mailboxes
       (``parallel.ipc.close_all``: peers' mappings closed, own memory freed);
    4. ``destroy_process_group()`` (c10d's own communicators and the store).

    If a peer died and a step blocks, the timer ends this process with exit code 3 after ``timeout_s`` instead of
    holding the launcher (which tears the group down on the first non-zero exit)."""
    import threading

    done = threading.Event()

    def _bail():
        if not done.wait(timeout_s):
            import sys

            print(f"[pdt] shutdown did not finish within {timeout_s:.0f} s; exiting", file=sys.stderr, flush=True)
            os._exit(3)

    t = threading.Thread(target=_bail, name="pdt-shutdown-timer", daemon=True)
    t.start()
    try:
        barrier()
        if engine is not None:
            engine.close()
        try:
            from ..parallel.reducer_native import close_all_comms

            close_all_comms()
        except ImportError:
            pass
        from ..parallel import ipc

        ipc.close_all()  # hipIpc mappings of the peers' mailboxes closed, own mailboxes freed
        destroy_process_group()
    finally:
        done.set()


def staged_through_host(tensor: torch.Tensor, group=None) -> bool:
    """GPU tensor on a group whose backend is not RCCL (gloo: the one-GPU rehearsal of W > 1, or a GPU job forced onto
    ``--backend gloo``).  c10d's gloo backend would move it through the host on its OWN pool streams and worker threads
    (a worker thread blocks in ``hipStreamSynchronize`` while the caller blocks in ``work.wait()``); with more HIP
    streams than hardware queues (``GPU_MAX_HW_QUEUES``, 4 by default) two ranks sharing a GPU stopped in backward in
    2 runs of 2, and in 0 of 4 with 16 queues (``profiles/r3_w2_hang.md``).  Such tensors are therefore staged by US:
    a synchronous copy to the host on the current stream, the collective on the CPU tensor, a copy back.  No pool
    stream, no worker thread that waits for the device."""
    return tensor.is_cuda and tdist.get_backend(group) != "nccl"


def all_reduce_(tensor: torch.Tensor, group=None, op=None) -> torch.Tensor:
    """``tdist.all_reduce`` in place, host-staged where :func:`staged_through_host` says so."""
    op = tdist.ReduceOp.SUM if op is None else op
    if staged_through_host(tensor, group):
        h = tensor.detach().cpu()
        tdist.all_reduce(h, op=op, group=group)
        tensor.copy_(h)
    else:
        tdist.all_reduce(tensor, op=op, group=group)
    return tensor


def broadcast_(tensor: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    if staged_through_host(tensor, group):
        h = tensor.detach().cpu()
        tdist.broadcast(h, src=src, group=group)
        tensor.copy_(h)
    else:
        tdist.broadcast(tensor, src=src, group=group)
    return tensor


def reduce_loss(tensor: torch.Tensor, rank: int, world_size: int) -> None:
    """``main.py:16-20``: SUM-reduce to rank 0, rank 0 divides by world size.

    In place, under ``no_grad``; non-destination ranks keep whatever the
    backend leaves in their buffer (their local loss for RCCL and gloo).
    """
    with torch.no_grad():
        if is_initialized() and world_size > 1:
            if staged_through_host(tensor):
                h = tensor.detach().cpu()
                tdist.reduce(h, dst=0)
                if rank == 0:  # (the other ranks keep their local loss, as with the device path)
                    tensor.copy_(h)
            else:
                tdist.reduce(tensor, dst=0)
        if rank == 0:
            tensor /= world_size


def all_reduce_mean_(tensor: torch.Tensor) -> torch.Tensor:
    """Mean over ranks, in place (used by the synchronised metrics path)."""
    if is_initialized() and get_world_size() > 1:
        all_reduce_(tensor)
        tensor /= get_world_size()
    return tensor
