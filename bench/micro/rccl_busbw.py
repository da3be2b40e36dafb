"""All-reduce calibration for the gradient path: bus bandwidth and per-collective latency on THIS node's links.

    python bench/micro/rccl_busbw.py --gpus 8                       # RCCL over xGMI, sweep of NCCL_ALGO x NCCL_PROTO
    python bench/micro/rccl_busbw.py --gpus 2 --backend gloo --device cpu --sizes_mb 1 4   # logic check without GPUs

The launcher starts one job per (NCCL_ALGO, NCCL_PROTO) combination (the variables are read when the communicator is
created, so each combination needs fresh processes), every job times ``all_reduce`` for 4..64 MiB buffers and for a
4 KiB one (latency), and the launcher prints a table plus one JSON document:

    {"world": 8, "best": {"algo": "Ring", "proto": "Simple"}, "link_GBps": ..., "collective_latency_us": ...,
     "bucket_mb": ..., "rows": [...]}

``link_GBps`` is the per-link figure the ring model of ``bench/scaling.py`` wants (bus bandwidth: algorithm bandwidth
x 2(N-1)/N), ``bucket_mb`` the smallest size within 10 % of the best bus bandwidth (what ``DistributedDataParallel(
bucket_cap_mb="auto")`` would pick), so ``bench/scaling.py --calibration <file>`` replaces its guessed constants
(153 GB/s, 25 us: the task brief's numbers, never measured on this pool, which exposes one GPU per job).

Reference: the stock trainer relies on NCCL's defaults (/root/reference/main.py:83); SURVEY 5.8 layer 4.
"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(a) -> int:
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if a.device == "cuda":
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")
    dist.init_process_group(a.backend, init_method="env://")

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    rows = []
    for mb in [4.0 / 1024] + list(a.sizes_mb):
        n = max(1, int(mb * (1 << 20)) // 4)
        t = torch.ones(n, dtype=torch.float32, device=dev)
        for _ in range(a.warmup):
            dist.all_reduce(t)
        sync()
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            dist.all_reduce(t)
        sync()
        dt = torch.tensor([(time.perf_counter() - t0) / a.iters], dtype=torch.float64, device=dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        sec = float(dt.item())
        algbw = n * 4 / sec / 1e9
        rows.append({"mb": round(mb, 4), "us": round(sec * 1e6, 2), "algbw_GBps": round(algbw, 3),
                     "busbw_GBps": round(algbw * 2 * (world - 1) / max(world, 1), 3)})
    ipc_rows = None
    if a.ipc and dev.type == "cuda" and world > 1:
        # the framework's own two-shot direct all-reduce over IPC windows (csrc/comm/ipc_allreduce.hip) on the same sizes:
        # PDT_GRAD_COMM=auto makes this comparison at start-up per job; here it is a table
        sys.path.insert(0, ROOT)
        from pytorch_distributed_training_amd.parallel.ipc import IpcBucketAllReduce

        red = IpcBucketAllReduce(None, dev, int(max(a.sizes_mb) * (1 << 20)))
        if red.ok:
            ipc_rows = []
            for mb in [4.0 / 1024] + list(a.sizes_mb):
                n = max(2, int(mb * (1 << 20)) // 4)
                t = torch.ones(n, dtype=torch.float32, device=dev)
                for _ in range(a.warmup):
                    red.all_reduce_(t)
                sync()
                dist.barrier()
                sync()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    red.all_reduce_(t)
                sync()
                dt = torch.tensor([(time.perf_counter() - t0) / a.iters], dtype=torch.float64, device=dev)
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
                sec = float(dt.item())
                algbw = n * 4 / sec / 1e9
                ipc_rows.append({"mb": round(mb, 4), "us": round(sec * 1e6, 2), "algbw_GBps": round(algbw, 3),
                                 "busbw_GBps": round(algbw * 2 * (world - 1) / max(world, 1), 3)})
            sync()
            dist.barrier()
            red.close()
        elif rank == 0:
            print(f"two-shot IPC all-reduce unavailable: {red.error}", file=sys.stderr, flush=True)
    if rank == 0:
        print("RESULT " + json.dumps({"world": world, "algo": os.environ.get("NCCL_ALGO", "default"),
                                      "proto": os.environ.get("NCCL_PROTO", "default"), "rows": rows,
                                      "two_shot_ipc_rows": ipc_rows}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def summarize(results: list[dict]) -> dict:
    """Best combination by bus bandwidth at the largest size; latency from the 4 KiB row of that combination."""
    ok = [r for r in results if r.get("rows")]
    if not ok:
        return {"error": "no combination completed", "rows": results}
    best = max(ok, key=lambda r: r["rows"][-1]["busbw_GBps"])
    big = [row for row in best["rows"] if row["mb"] >= 1.0] or best["rows"][-1:]
    top = max(row["busbw_GBps"] for row in big)
    bucket = min(row["mb"] for row in big if row["busbw_GBps"] >= 0.9 * top)
    return {"world": best["world"], "best": {"algo": best["algo"], "proto": best["proto"]},
            "link_GBps": top, "collective_latency_us": best["rows"][0]["us"], "bucket_mb": bucket, "rows": results}


def launch(a) -> int:
    algos = a.algos if a.backend == "nccl" else ["default"]
    protos = a.protos if a.backend == "nccl" else ["default"]
    results = []
    for i, (algo, proto) in enumerate(itertools.product(algos, protos)):
        env = dict(os.environ, PYTHONPATH=ROOT)
        if algo != "default":
            env["NCCL_ALGO"] = algo
        if proto != "default":
            env["NCCL_PROTO"] = proto
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(a.port + i), os.path.abspath(__file__), "--worker",
               "--backend", a.backend, "--device", a.device, "--iters", str(a.iters), "--warmup", str(a.warmup)] + \
              (["--ipc"] if (a.ipc and i == 0) else []) + ["--sizes_mb"] + [str(s) for s in a.sizes_mb]
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=a.timeout)
        except subprocess.TimeoutExpired:
            results.append({"algo": algo, "proto": proto, "error": "timeout"})
            continue
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if out.returncode != 0 or not line:  # (RCCL refuses some ALGO x PROTO pairs: recorded, not fatal)
            results.append({"algo": algo, "proto": proto, "error": (out.stderr or out.stdout)[-400:]})
            continue
        results.append(json.loads(line[-1][len("RESULT "):]))
    print(f"{'algo':>8} {'proto':>8} " + " ".join(f"{s:>9g}MiB" for s in a.sizes_mb) + "   4KiB us   (bus GB/s)")
    for r in results:
        if "rows" not in r:
            print(f"{r['algo']:>8} {r['proto']:>8}  failed: {r['error'].strip().splitlines()[-1] if r['error'].strip() else ''}")
            continue
        print(f"{r['algo']:>8} {r['proto']:>8} " + " ".join(f"{row['busbw_GBps']:>12.1f}" for row in r["rows"][1:])
              + f"   {r['rows'][0]['us']:8.1f}")
    ipc = next((r["two_shot_ipc_rows"] for r in results if r.get("two_shot_ipc_rows")), None)
    if ipc:
        print(f"{'two-shot':>8} {'ipc':>8} " + " ".join(f"{row['busbw_GBps']:>12.1f}" for row in ipc[1:]) + f"   {ipc[0]['us']:8.1f}")
    doc = summarize(results)
    doc["two_shot_ipc_rows"] = ipc
    text = json.dumps(doc)
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    return 0 if "error" not in doc else 1


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2, help="ranks (one per GPU with --device cuda)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--sizes_mb", type=float, nargs="+", default=[4, 8, 16, 32, 64])
    ap.add_argument("--algos", nargs="+", default=["default", "Ring", "Tree"])
    ap.add_argument("--protos", nargs="+", default=["default", "Simple", "LL", "LL128"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--port", type=int, default=29710)
    ap.add_argument("--timeout", type=int, default=300)
    ap.add_argument("--out", default=None)
    ap.add_argument("--ipc", action="store_true", help="also time the framework's two-shot direct all-reduce over IPC windows")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    return worker(a) if a.worker else launch(a)


if __name__ == "__main__":
    sys.exit(main())
