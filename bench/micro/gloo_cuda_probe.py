"""Does this build's gloo backend all-reduce GPU tensors?  (two ranks sharing GPU 0; run by hand on the GPU box)"""
import os
import sys

import torch
import torch.distributed as tdist

tdist.init_process_group("gloo")
r = tdist.get_rank()
torch.cuda.set_device(0)
x = torch.full((1000,), float(r + 1), device="cuda")
try:
    tdist.all_reduce(x)
    torch.cuda.synchronize()
    print(f"rank {r}: gloo all_reduce on a GPU tensor -> {float(x[0])}", flush=True)
except Exception as e:
    print(f"rank {r}: gloo refuses GPU tensors: {e!r}", flush=True)
tdist.barrier()
tdist.destroy_process_group()
