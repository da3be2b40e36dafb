"""One rank of the two-shot direct all-reduce test (tests/test_gpu.py): W processes share GPU 0 behind a gloo group and
all-reduce bucket-sized fp32 / bf16 buffers through each other's IPC windows (csrc/comm/ipc_allreduce.hip); sums are
checked against a CPU all-reduce, bits against the other ranks.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    import faulthandler

    faulthandler.dump_traceback_later(150, exit=True)
    from pytorch_distributed_training_amd.parallel import ipc as pipc

    tdist.init_process_group("gloo")
    rank, world = tdist.get_rank(), tdist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {"rank": rank, "world": world}
    bucket_bytes = 4 << 20
    red = pipc.IpcBucketAllReduce(None, dev, bucket_bytes, timeout_s=20.0)
    out["setup_error"] = red.error
    if not red.ok:
        print(json.dumps(out), flush=True)
        return 0
    out["capacity_words"] = int(red.native.large_capacity_words)

    gen = torch.Generator().manual_seed(7 + rank)
    worst = {"fp32": 0.0, "bf16": 0.0}
    same_bits = True
    for dtype, key in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        for n in (2, 6, 1000, 4098, 65542, 1 << 20):
            for rep in range(2):
                x = torch.randn(n, generator=gen).to(dtype)
                want = x.float().clone()
                tdist.all_reduce(want)  # fp32 sum of the (already rounded) inputs
                got = red.all_reduce_(x.to(dev)).cpu()
                err = float((got.float() - want).abs().max() / want.abs().max().clamp_min(1e-6))
                worst[key] = max(worst[key], err)
                everyone = [torch.empty_like(got) for _ in range(world)]
                tdist.all_gather(everyone, got)
                same_bits = same_bits and all(torch.equal(e, everyone[0]) for e in everyone)
    out["worst_rel_err"], out["same_bits"] = worst, same_bits

    # fp32: the kernel adds in rank order, exactly like a sequential sum over the ranks
    x = torch.randn(4099, generator=gen)
    parts = [torch.empty_like(x) for _ in range(world)]
    tdist.all_gather(parts, x)
    seq = parts[0].clone()
    for p in parts[1:]:
        seq = seq + p
    got = red.all_reduce_(x.to(dev)).cpu()
    out["rank_order_exact"] = bool(torch.equal(got, seq))

    err, t_ipc, t_pg = red.validate(bucket_bytes // 4, check_timeout_s=10.0)
    out["validate_error"], out["ipc_ms"], out["group_ms"] = err, t_ipc, t_pg

    # replayable inside a hipGraph
    buf = torch.full((70000,), float(rank + 1), device=dev)
    torch.cuda.synchronize()
    tdist.barrier()
    s = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        red.all_reduce_(buf)
    vals = []
    for _ in range(3):
        buf.fill_(float(rank + 1))
        g.replay()
        torch.cuda.synchronize()
        vals.append([float(buf[0]), float(buf[-1]), float(buf.min()), float(buf.max())])
    out["graph_vals"] = vals
    out["graph_want"] = float(sum(range(1, world + 1)))

    # a bucket beyond the windows is refused, not truncated
    try:
        red.all_reduce_(torch.zeros(out["capacity_words"] + 2, device=dev))
        out["refused_oversize"] = False
    except RuntimeError:
        out["refused_oversize"] = True
    torch.cuda.synchronize()
    tdist.barrier()
    red.close()
    print(json.dumps(out), flush=True)
    tdist.barrier()
    tdist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
