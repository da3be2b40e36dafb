"""One rank of the IPC small-message all-reduce test (tests/test_gpu.py): W processes that all use GPU 0, a gloo
group for the bootstrap, ``_C.IpcAllReduce`` mailboxes mapped across the processes with hipIpc handles.
Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    from pytorch_distributed_training_amd import nn as pnn
    from pytorch_distributed_training_amd.parallel import ipc as pipc

    tdist.init_process_group("gloo")
    rank, world = tdist.get_rank(), tdist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {"rank": rank}

    red = pipc.IpcSmallAllReduce(None, dev, timeout_s=10.0)
    out["setup_error"] = red.error
    if not red.ok:
        print(json.dumps(out), flush=True)
        return 0

    # 1. sums against a CPU all-reduce, several calls back to back (both mailbox sets), identical bits on all ranks
    gen = torch.Generator().manual_seed(100 + rank)
    worst, same_bits = 0.0, True
    for n in (1, 63, 132, 1028, 4100, 16384):
        for _ in range(4):
            x = torch.randn(n, generator=gen)
            want = x.clone()
            tdist.all_reduce(want)
            got = red.all_reduce_(x.to(dev)).cpu()
            worst = max(worst, float((got - want).abs().max() / want.abs().max().clamp_min(1e-6)))
            everyone = [torch.empty_like(got) for _ in range(world)]
            tdist.all_gather(everyone, got)
            same_bits = same_bits and all(torch.equal(e, everyone[0]) for e in everyone)
    out["worst_rel_err"], out["same_bits"] = worst, same_bits

    # 2. the self-check + timing that decides the path in a job
    err, t_ipc, t_pg = red.validate(check_timeout_s=5.0)
    out["validate_error"], out["ipc_us"], out["group_us"] = err, t_ipc, t_pg

    # 3. replayable inside a hipGraph (the sequence number lives in device memory)
    buf = torch.full((256,), float(rank + 1), device=dev)
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
        vals.append(float(buf[0]))
    out["graph_vals"] = vals
    out["graph_want"] = float(sum(range(1, world + 1)))

    # 4. SyncBatchNorm through the IPC path == plain BN over the concatenated batch
    os.environ["PDT_SYNCBN_COMM"] = "ipc"
    os.environ["PDT_QUIET"] = "1"
    torch.manual_seed(0)
    c, per = 64, 8
    full = torch.randn(per * world, c, 8, 8).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dfull = torch.randn(per * world, c, 8, 8).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bn_ref = pnn.BatchNorm2d(c).to(dev)
    bn_sync = pnn.SyncBatchNorm(c).to(dev)
    with torch.no_grad():
        w = torch.rand(c, device=dev) + 0.5
        b = torch.randn(c, device=dev)
        for m in (bn_ref, bn_sync):
            m.weight.copy_(w)
            m.bias.copy_(b)
    xr = full.clone().requires_grad_(True)
    yr = bn_ref(xr, relu=True)
    yr.backward(dfull)
    xs = full[rank * per:(rank + 1) * per].clone().requires_grad_(True)
    ys = bn_sync(xs, relu=True)
    ys.backward(dfull[rank * per:(rank + 1) * per])
    sl = slice(rank * per, (rank + 1) * per)
    out["syncbn_path"] = pipc.small_all_reduce_path(None, dev)[1]["path"]
    out["syncbn_fwd_err"] = float((ys.float() - yr[sl].float()).abs().max())
    out["syncbn_dx_err"] = float((xs.grad.float() - xr.grad[sl].float()).abs().max() / xr.grad.float().abs().max())
    out["syncbn_rm_err"] = float((bn_sync.running_mean - bn_ref.running_mean).abs().max())
    out["syncbn_rv_err"] = float((bn_sync.running_var - bn_ref.running_var).abs().max())

    # 4b. the exchange happens INSIDE the BN kernels: a SyncBN forward + backward launches exactly the kernels of a
    # plain BN forward + backward (no packing, all-reduce, clone or fill launches around them); small tensor (the
    # backward exchange runs in the prologue of the grad-input kernel) and a large one (in the coefficient kernel)
    from torch.profiler import ProfilerActivity, profile

    def kernels_of(fn):
        torch.cuda.synchronize()
        tdist.barrier()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        names = []
        for e in prof.events():
            if str(e.device_type).endswith("CUDA") and "memcpy" not in e.name.lower() and "memset" not in e.name.lower():
                names.append(e.name.split("(")[0].split("<")[0].split("::")[-1])
        return names

    out["kernel_names"] = {}
    for tag, shape in (("small", (per, c, 8, 8)), ("large", (64, c, 112, 112))):
        xa = torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        da = torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

        def step(mod):
            xi = xa.clone().requires_grad_(True)
            mod(xi, relu=True).backward(da)

        step(bn_ref), step(bn_sync)  # warm-up
        k_ref = kernels_of(lambda: step(bn_ref))
        k_sync = kernels_of(lambda: step(bn_sync))
        out["kernel_names"][tag] = [k_ref, k_sync]

    # 4d. cost of a SyncBN layer (forward + backward, CIFAR-sized ResNet-50 activations, 64 images per rank): plain BN,
    # SyncBN with the exchange inside the BN kernels, SyncBN through the packed tensor + separate all-reduce kernel
    import time

    layers = [(64, 32), (256, 32), (128, 16), (512, 16), (256, 8), (1024, 8), (512, 4), (2048, 4)]
    mods = []
    for ch, hw in layers:
        xa = torch.randn(64, ch, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        mods.append((pnn.BatchNorm2d(ch).to(dev), pnn.SyncBatchNorm(ch).to(dev), xa, torch.randn_like(xa)))

    def sweep(which):
        for ref_m, sync_m, xa, da in mods:
            xi = xa.clone().requires_grad_(True)
            (ref_m if which == "bn" else sync_m)(xi, relu=True).backward(da)

    out["layer_us"] = {}
    for which, ink in (("bn", "1"), ("syncbn_inkernel", "1"), ("syncbn_packed", "0")):
        os.environ["PDT_SYNCBN_INKERNEL"] = ink
        sweep(which)
        torch.cuda.synchronize()
        tdist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            sweep(which)
        torch.cuda.synchronize()
        out["layer_us"][which] = (time.perf_counter() - t0) / 20 / len(layers) * 1e6
    os.environ["PDT_SYNCBN_INKERNEL"] = "1"

    # 4c. statistics with |mean| = 100 std and a different mean on every rank: Chan's merge of (count, mean, M2)
    # in the finalize kernel against a float64 computation over the concatenated batch
    gen2 = torch.Generator().manual_seed(7)
    big = torch.cat([(torch.randn(per, c, 8, 8, generator=gen2) + 100.0 + 3.0 * r) for r in range(world)])
    big = big.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref64 = big.double()
    var64 = ref64.var(dim=(0, 2, 3), unbiased=False)
    bn_big = pnn.SyncBatchNorm(c, momentum=1.0).to(dev)
    bn_big(big[rank * per:(rank + 1) * per], relu=False)
    out["syncbn_big_mean_err"] = float((bn_big.running_mean.double() - ref64.mean(dim=(0, 2, 3))).abs().max())
    n_tot = big.numel() / c
    out["syncbn_big_var_rel_err"] = float(((bn_big.running_var.double() * (n_tot - 1) / n_tot - var64) / var64).abs().max())

    # 4b. a bottleneck net with downsample branches under SyncBN: the block-output BN and the downsample BN share one
    # pass over their gradient (bn_bwd_elemt2) and BOTH exchange their sums inside the table kernels
    import copy

    import pytorch_distributed_training_amd.models.resnet as R
    from pytorch_distributed_training_amd.ops import native
    from pytorch_distributed_training_amd.parallel import ParamArena

    torch.manual_seed(11)
    net_ref = R.ResNet(R.Bottleneck, [1, 1, 1, 1], num_classes=10).to(dev)
    net_sync = pnn.convert_sync_batchnorm(copy.deepcopy(net_ref))
    arenas = [ParamArena(net_ref), ParamArena(net_sync)]
    for ar in arenas:
        ar.attach_lowp(torch.bfloat16)
    nb = 8
    gen3 = torch.Generator().manual_seed(13)
    xf = torch.randn(nb * world, 3, 64, 64, generator=gen3).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    sl2 = slice(rank * nb, (rank + 1) * nb)
    calls = []
    orig2 = native.bn_bwd_elemt2
    native.bn_bwd_elemt2 = lambda *a_, **k_: (calls.append(k_.get("ipc") is not None), orig2(*a_, **k_))[1]
    try:
        o_ref = net_ref(xf)
        o_ref.float().square().sum().backward()
        n_ref_calls = len(calls)
        o_sync = net_sync(xf[sl2].contiguous(memory_format=torch.channels_last))
        o_sync.float().square().sum().backward()
    finally:
        native.bn_bwd_elemt2 = orig2
    torch.cuda.synchronize()
    out["dual_calls"] = [n_ref_calls, len(calls) - n_ref_calls, all(calls[n_ref_calls:]), any(calls[:n_ref_calls])]
    out["dual_out_err"] = float((o_sync.float() - o_ref[sl2].float()).abs().max() / o_ref.float().abs().max())
    # weight gradients: the ranks' local gradients add up to the single-process ones (what DDP's all-reduce does)
    cos = 1.0
    for (n1, p1), (_, p2) in zip(net_ref.named_parameters(), net_sync.named_parameters()):
        if not (n1.endswith("conv1.weight") or n1.endswith("downsample.0.weight") or n1 == "fc.weight"):
            continue
        g2 = p2.grad.detach().float().cpu()
        tdist.all_reduce(g2)
        g1 = p1.grad.detach().float().cpu().flatten()
        cos = min(cos, float(torch.dot(g2.flatten(), g1) / (g2.norm() * g1.norm() + 1e-12)))
    out["dual_dw_cos"] = cos

    # 4e. a step captured in a hipGraph whose BN kernels do the exchange themselves (engine.CapturedStep): replays equal
    # the eager step on both ranks; then rank 1 stops replaying.  Rank 0's next replay runs out of time in its FIRST
    # exchange, the exchanges queued behind it inside the graph see the device-side flag and do not wait again, and
    # the replay after that raises on the host instead of returning a loss computed from stale statistics.
    from pytorch_distributed_training_amd.engine import CapturedStep

    path_ipc = pipc.small_all_reduce_path(None, dev)[0]
    limit_s = 2.0
    path_ipc.native.set_timeout(limit_s)  # (the time limit travels by value inside the captured kernel arguments)
    path_ipc.timeout_s = limit_s
    bn_g = pnn.SyncBatchNorm(c).to(dev)
    xg = full[rank * per:(rank + 1) * per].clone()

    def bn_step(xin):
        xi = xin.detach().clone().requires_grad_(True)
        o = bn_g(xi, relu=True).float()
        loss = (o * o).mean()
        loss.backward()
        return loss.detach()

    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        eager = [float(bn_step(xg)) for _ in range(2)]
    torch.cuda.synchronize()
    tdist.barrier()
    cs = CapturedStep(bn_step, (xg.clone(),), side, path_ipc)
    reps = []
    for _ in range(3):
        reps.append(float(cs.replay(xg)))
    torch.cuda.synchronize()
    out["captured_losses"] = [eager, reps]
    tdist.barrier()
    if rank == 0:
        t0 = time.perf_counter()
        cs.replay(xg)          # the peer is absent: forward exchange times out, backward exchange must not wait again
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        try:
            cs.replay(xg)
            raised = False
        except RuntimeError as e:
            raised = "peer did not arrive" in str(e)
        out["captured_peer_missing"] = {"elapsed_s": dt, "limit_s": limit_s, "raised": raised}
    tdist.barrier()

    # 5. a peer that never arrives: the kernel gives up after the limit, the flag is raised, the next call refuses
    torch.cuda.synchronize()
    tdist.barrier()
    if rank == 0:
        red.native.set_timeout(1.0)
        red.all_reduce_(torch.ones(64, device=dev))
        torch.cuda.synchronize()
        out["timed_out"] = red.timed_out()
        try:
            red.all_reduce_(torch.ones(64, device=dev))
            out["refused_after_timeout"] = False
        except RuntimeError:
            out["refused_after_timeout"] = True
    tdist.barrier()
    # 6. ordered teardown (dist.shutdown's step 3): every path closed in two phases around a barrier, nothing left
    n_paths = len([1 for v in pipc._paths.values() if v[0] is not None])
    out["closed_paths"] = [n_paths, pipc.close_all(), len(pipc._paths), path_ipc.native is None]
    red.close()
    out["red_closed"] = red.native is None and not red.ok
    print(json.dumps(out), flush=True)
    tdist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
