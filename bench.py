"""Headline benchmark: ResNet-50 bf16 DDP training throughput on MI355X.

    python bench.py --gpus 1 --steps 30 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): whole-job images/sec of ResNet-50 (v1.5, 1000 classes,
random init) training in bf16 with DDP on synthetic ImageNet-shaped data
(3x224x224), one process per GPU over RCCL, weak scaling (fixed per-GPU batch).
Every timed step is a full training iteration: input cast/layout, forward,
fused softmax-CE, backward with bucketed gradient all-reduce, fused SGD step.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# Published comparison points recorded in BASELINE.md (NVIDIA DLE ResNet-50 v1.5, DGX A100, AMP):
# 1 GPU ~2470 img/s, 8 GPUs ~16600 img/s. Intermediate N: linear interpolation of per-GPU throughput.
_BASE_1, _BASE_8 = 2470.0, 16600.0


def baseline_for(n_gpus: int) -> float:
    per1, per8 = _BASE_1, _BASE_8 / 8.0
    per = per1 + (per8 - per1) * (min(max(n_gpus, 1), 8) - 1) / 7.0
    return per * n_gpus


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    # per-GPU batch sized for 288 GB HBM3E: a ResNet-50 step at 224x224 holds ~41 MiB of bf16 activations per image
    # (105 GiB at the default 2560, measured peak).  Activations beyond one 2 GiB buffer descriptor are launched in
    # equal batch chunks; 2560 = two chunks of 1280 for the widest layers is the measured optimum (end of round 2:
    # 14.7k img/s at 1280, 14.8k at 2048, 15.0k at 2304, 15.1-15.2k at 2560, 14.8k at 3072, 14.6-14.7k at 4096 / 5120);
    # every conv launch has >12 rounds of tiles per CU and the gradient all-reduce is <1 % of the step.
    # Any other network / image size without --batch_size: the memory model's pick at the same ~40 % of the device.
    ap.add_argument("--batch_size", type=int, default=None,
                    help="per-GPU batch (default: 2560 for ResNet-50 at 224x224, else the memory model's pick at 40 %% "
                         "of the device, at most 2560); 0 = the memory model's pick at 85 %% (utils/memory.py)")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--image_size", type=int, default=224)
    ap.add_argument("--num_classes", type=int, default=1000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "adam", "adamw"])
    ap.add_argument("--sync_bn", action="store_true")
    ap.add_argument("--bucket_cap_mb", type=lambda v: v if v == "auto" else float(v), default=None)
    ap.add_argument("--grad_dtype", default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient buckets")
    ap.add_argument("--grad_comm", default=None, choices=["rccl", "ipc", "auto"],
                    help="gradient buckets through ncclAllReduce (default), the two-shot IPC kernel, or the faster of the two")
    ap.add_argument("--backend", default=None)
    ap.add_argument("--device", default=None, choices=["auto", "cuda", "cpu"],
                    help="with --backend gloo: cuda keeps the tensors on the GPU (several ranks may share one)")
    ap.add_argument("--reducer", default=None)
    ap.add_argument("--graph", action="store_true", help="capture the steady-state step in one hipGraph")
    args = ap.parse_args(argv)
    if args.grad_comm is not None:
        os.environ["PDT_GRAD_COMM"] = args.grad_comm  # (read when the reducer is built)
    # stdout carries exactly ONE line, the JSON result: everything else this process or its libraries print
    # (RCCL's version banner, c10d notices) goes to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    from pytorch_distributed_training_amd import dist as pdist
    from pytorch_distributed_training_amd import native_available
    from pytorch_distributed_training_amd.data import SyntheticDeviceLoader
    from pytorch_distributed_training_amd.engine import Engine, EngineConfig
    from pytorch_distributed_training_amd.nn import functional as pF

    # N = 1 times the SAME path as N > 1: a size-1 RCCL communicator makes the reducer run flatten -> event fence ->
    # ncclAllReduce on the communicator's stream -> completion events per bucket, instead of skipping the collective
    # (the driver's scaling efficiency divides by this number).  PDT_FORCE_COMM=0 restores the communicator-less step.
    os.environ.setdefault("PDT_FORCE_COMM", "1")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus and world_env > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    env = pdist.init_process_group(backend=args.backend, local_rank=int(os.environ.get("LOCAL_RANK", "0")),
                                   device=args.device)
    on_gpu = env.device.type == "cuda"
    if on_gpu and not native_available():
        raise SystemExit("the native extension is not built: run `python __graft_entry__.py build`")

    engine = Engine(EngineConfig(
        model=args.model, num_classes=args.num_classes, dtype=args.dtype, optimizer=args.optimizer,
        lr=0.1 if args.optimizer == "sgd" else 1e-3, momentum=0.9, weight_decay=1e-4 if args.optimizer == "sgd" else 0.0,
        sync_bn=args.sync_bn, bucket_cap_mb=args.bucket_cap_mb, grad_dtype=args.grad_dtype, reducer=args.reducer, seed=0, graph=args.graph,
        image_size=args.image_size))
    batch_model = None
    flagship = args.model == "resnet50" and args.image_size == 224
    if args.batch_size is None and (flagship or not on_gpu):
        args.batch_size = 2560 if on_gpu else 32
    if args.batch_size is None or args.batch_size <= 0:
        from pytorch_distributed_training_amd.utils.memory import suggest_batch

        shape = (3, args.image_size, args.image_size)
        if args.batch_size is None:  # a deeper / wider network at 2560 images may not fit: same share of the device
            batch_model = suggest_batch(engine, shape, args.num_classes, fraction=0.4, cap=2560)
        else:
            batch_model = suggest_batch(engine, shape, args.num_classes)
        args.batch_size = batch_model["batch"]
    loader = SyntheticDeviceLoader(args.batch_size, (3, args.image_size, args.image_size), args.num_classes,
                                   device=env.device, dtype=torch.float32, ring=2, seed=env.rank)
    it = iter(loader)

    def sync():
        pdist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    loss = None
    # The first iterations are not steady state whatever the caller's --warmup is: lazy code-object loading,
    # allocator growth, and the reducer's bucket rebuild at the start of iteration 2.  With fewer than 3 warm-up
    # steps requested, the missing ones run here, untimed and reported as config.priming_steps.
    priming = max(0, 3 - args.warmup)
    for _ in range(priming + args.warmup):
        loss = engine.train_step(*next(it))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = engine.train_step(*next(it))
    sync()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=env.device if on_gpu else "cpu")
    if env.world_size > 1:
        import torch.distributed as tdist

        if on_gpu:
            t = t.float()
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    elapsed = float(t.item())
    final_loss = float(loss.item()) if loss is not None else float("nan")

    if env.rank == 0:
        n = env.world_size
        global_batch = args.batch_size * n
        value = global_batch * args.steps / elapsed
        base = baseline_for(n)
        out = {
            "metric": "images/sec (whole job) ResNet-50 bf16 DDP training" if args.model == "resnet50"
            else f"images/sec (whole job) {args.model} DDP training",
            "value": round(value, 2),
            "unit": "images/sec",
            "n_gpus": n,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": round(value / base, 4) if args.model == "resnet50" else None,
            "dtype": args.dtype if on_gpu else "fp32",
            "data": "synthetic",
            "config": {
                "model": args.model, "global_batch": global_batch, "per_gpu_batch": args.batch_size,
                "image_size": args.image_size, "seq_len": None, "parallelism": f"dp{n}",
                "optimizer": args.optimizer, "sync_bn": bool(args.sync_bn), "reducer": engine.net.reducer_kind,
                "backend": env.backend, "baseline_images_per_sec": round(base, 1),
                "library_fallbacks": dict(pF.fallback_counts),
                "grad_comm": getattr(engine.net.reducer, "bucket_comm", None), "final_loss": final_loss,
                "hip_graph": bool(engine._graph is not None), "batch_model": batch_model, "bucket_autotune": engine.net.bucket_autotune,
                "priming_steps": priming,
                # True: every bucket went through ncclAllReduce on the own communicator (also at N = 1)
                "native_comm": getattr(engine.net.reducer, "rccl", None) is not None,
                "sync_bn_comm": engine.sync_bn_comm, "grad_dtype": args.grad_dtype,
            },
        }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    sys.stdout.flush()
    # ordered teardown (barrier -> reducer hooks off -> communicator drained and destroyed -> c10d), bounded by a timer
    pdist.shutdown(engine, timeout_s=60.0)
    return 0


if __name__ == "__main__":
    sys.exit(main())
