"""Path selection of the SyncBN statistics all-reduce (parallel/ipc.py) where no GPU is involved: CPU tensors
always take the process group, and the switch PDT_SYNCBN_COMM=torch never touches the IPC set-up.  The IPC kernel
itself is covered on the GPU (tests/test_gpu.py::test_ipc_small_allreduce_two_processes_on_one_gpu)."""
import os

import torch
import torch.distributed as tdist

from pytorch_distributed_training_amd.parallel import ipc as pipc


def _single_rank_group(tmp_path):
    if not tdist.is_initialized():
        tdist.init_process_group("gloo", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1)


def test_cpu_tensors_use_the_process_group(tmp_path):
    _single_rank_group(tmp_path)
    try:
        pipc.reset_paths()
        os.environ["PDT_QUIET"] = "1"
        red, info = pipc.small_all_reduce_path(None, torch.device("cpu"))
        assert red is None and info["path"] == "group" and info["why"] == "cpu tensors"
        t = torch.arange(8, dtype=torch.float32)
        assert torch.equal(pipc.small_all_reduce(t.clone()), t)
        # cached: the decision is made once per (group, device)
        assert pipc.small_all_reduce_path(None, torch.device("cpu"))[1] is info
    finally:
        pipc.reset_paths()
        os.environ.pop("PDT_QUIET", None)
        tdist.destroy_process_group()


def test_switch_selects_the_process_group_without_touching_ipc(monkeypatch):
    monkeypatch.setenv("PDT_SYNCBN_COMM", "torch")

    def boom(*a, **k):
        raise AssertionError("IPC set-up must not run")

    monkeypatch.setattr(pipc, "IpcSmallAllReduce", boom)
    red, info = pipc._decide(None, torch.device("cuda", 0))
    assert red is None and info == {"mode": "torch", "path": "group", "why": "selected", "ipc_us": None,
                                    "group_us": None}
