#!/usr/bin/env python
"""Does the native bf16 path LEARN like the fp32 oracle does?

Trains the same model from the same initial weights on the same stream of synthetic, learnable images
(class prototype + noise; there is no dataset on the box) through three op back ends, each in its own process
because the back end is chosen at import time:

  native     the HIP kernels of this package, bf16 compute, fp32 master weights       (what bench.py times)
  plain      stock PyTorch-ROCm ops (MIOpen / ATen), bf16 compute                     (PDT_PLAIN_TORCH=1)
  oracle     the fp32 PyTorch reference ops                                           (PDT_FORCE_REFERENCE_OPS=1, fp32)

and prints the training-loss curve (mean over windows of steps) and the held-out top-1 accuracy of each.  The
reference's script trains ResNet-50 + SyncBN with Adam(lr 1e-3) on CIFAR-shaped input (/root/reference/main.py:40-84);
that is the default configuration here.

  python bench/convergence.py                       # all three back ends, one summary table
  python bench/convergence.py --backend native      # one back end, one JSON line
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BACKENDS = {
    "native": ({}, "bf16"),
    "plain": ({"PDT_PLAIN_TORCH": "1"}, "bf16"),
    "oracle": ({"PDT_FORCE_REFERENCE_OPS": "1"}, "fp32"),
}


def make_batches(num_classes, image_size, seed):
    """Class prototypes (8x8 random fields, bilinearly upsampled) and a sampler of noisy views of them."""
    import torch

    g = torch.Generator().manual_seed(seed)
    low = torch.randn(num_classes, 3, 8, 8, generator=g)
    proto = torch.nn.functional.interpolate(low, size=(image_size, image_size), mode="bilinear", align_corners=False)

    def sample(n, gen, noise):
        y = torch.randint(0, num_classes, (n,), generator=gen)
        x = 0.5 + 0.2 * proto[y] + noise * torch.randn(n, 3, image_size, image_size, generator=gen)
        return x.clamp_(0.0, 1.0), y

    return sample


def run_one(args) -> dict:
    import torch

    from pytorch_distributed_training_amd import dist as pdist
    from pytorch_distributed_training_amd.engine import Engine, EngineConfig

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    gpu = torch.cuda.is_available()
    pdist.init_process_group(backend="rccl" if gpu else "gloo", local_rank=0)
    dtype = BACKENDS[args.backend][1] if gpu else "fp32"
    eng = Engine(EngineConfig(model=args.model, num_classes=args.num_classes, dtype=dtype, optimizer=args.optimizer,
                              lr=args.lr, sync_bn=True, seed=args.seed, image_size=args.image_size,
                              graph=bool(args.graph) and gpu, grad_dtype=args.grad_dtype))
    sample = make_batches(args.num_classes, args.image_size, seed=1)
    gen = torch.Generator().manual_seed(2 + 10 * args.seed)
    held = torch.Generator().manual_seed(3)
    eval_set = [sample(args.batch_size, held, args.noise) for _ in range(args.eval_batches)]
    losses, t0 = [], time.time()
    for step in range(args.steps):
        x, y = sample(args.batch_size, gen, args.noise)
        losses.append(eng.train_step(x.to(eng.device), y.to(eng.device)).clone())  # (a graph replay returns its static buffer)
        if (step + 1) % args.window == 0:
            print(f"[{args.backend}] step {step + 1}: loss {float(torch.stack(losses[-args.window:]).mean()):.4f}",
                  file=sys.stderr, flush=True)
    losses = [float(v) for v in torch.stack(losses).cpu()]
    acc = eng.evaluate(eval_set)
    curve = [sum(losses[i:i + args.window]) / len(losses[i:i + args.window]) for i in range(0, len(losses), args.window)]
    out = {"backend": args.backend, "dtype": dtype, "model": args.model, "image_size": args.image_size,
           "batch_size": args.batch_size, "steps": args.steps, "optimizer": args.optimizer, "lr": args.lr, "seed": args.seed, "graph": bool(eng._graph is not None), "grad_dtype": args.grad_dtype,
           "first_loss": losses[0], "loss_curve": [round(v, 4) for v in curve], "eval_top1": round(acc, 4),
           "finite": all(v == v and abs(v) != float("inf") for v in losses), "seconds": round(time.time() - t0, 1)}
    eng.close()
    pdist.shutdown()
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=list(BACKENDS), default=None)
    ap.add_argument("--backends", default="native,plain,oracle")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--image_size", type=int, default=32)
    ap.add_argument("--num_classes", type=int, default=100)
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--window", type=int, default=25)
    ap.add_argument("--eval_batches", type=int, default=16)
    ap.add_argument("--optimizer", default="adam")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--noise", type=float, default=0.25)
    ap.add_argument("--graph", type=int, default=0, help="1: replay the step from a hipGraph (what main.py does by default)")
    ap.add_argument("--grad_dtype", default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient buckets")
    ap.add_argument("--seed", type=int, default=0, help="initial weights and order of the training samples")
    ap.add_argument("--timeout", type=int, default=500)
    args = ap.parse_args()
    if args.backend is not None:
        print(json.dumps(run_one(args)), flush=True)
        return 0
    rows = []
    passthrough = []
    for k, v in vars(args).items():
        if k not in ("backend", "backends", "timeout"):
            passthrough += [f"--{k}", str(v)]
    for name in args.backends.split(","):
        env = dict(os.environ, **BACKENDS[name][0])
        r = subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "--backend", name] + passthrough, env=env,
                           stdout=subprocess.PIPE, text=True, timeout=args.timeout)
        if r.returncode != 0:
            print(f"[{name}] failed with exit code {r.returncode}", flush=True)
            return 1
        rows.append(json.loads(r.stdout.strip().splitlines()[-1]))
        print(json.dumps(rows[-1]), flush=True)
    print(f"\n| back end | dtype | first loss | loss, last {args.window} steps | held-out top-1 | s |\n|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['backend']} | {r['dtype']} | {r['first_loss']:.4f} | {r['loss_curve'][-1]:.4f} | {r['eval_top1']:.4f} | {r['seconds']} |")
    return 0


if __name__ == "__main__":
    sys.exit(main())
