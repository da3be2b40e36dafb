"""Scaling benchmark: ResNet-50 bf16 DDP images/sec at 1/2/4/8 MI355X.

    python bench/scaling.py --gpus 1 2 4 8 --steps 30 --warmup 10      # on a real multi-GPU node
    python bench/scaling.py --model-from gpurun_out/bench_native_b256.log  # modelled curve from a 1-GPU run

Measured mode launches ``bench.py`` through OUR launcher (one process per GPU, RCCL over xGMI) for
each N and reports whole-job images/sec and weak-scaling efficiency ``T(N) / (N * T(1))``.

Model mode (this build environment exposes one GPU per call) predicts the curve from the measured
1-GPU step time and the xGMI cost model of SURVEY §5.8: gradient all-reduce bandwidth term with the
per-collective latency, overlapped with backward except for the last bucket; plus the SyncBN
latency term (2 x 53 tiny all-reduces that cannot overlap) when ``--sync_bn``.  Every modelled
number is labelled as such.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Defaults from the task brief, NOT measured on this pool (it exposes one GPU per job): `--calibration FILE` replaces
# them with what bench/micro/rccl_busbw.py measured on the node at hand (bus bandwidth, latency, bucket size).
LINK_GBPS = 153.0          # xGMI per-link bandwidth
COLL_LATENCY_US = 25.0     # launch + sync latency per RCCL collective (10-30 us)
IPC_SMALL_US = 5.0         # SyncBN statistics exchange inside the BN kernels: +9.7 us per layer for forward + backward
                           # with 2 processes on one MI355X (profiles/r2_syncbn_inkernel.md); across xGMI add the
                           # store round trip, a few us
GRAD_BYTES = 25_557_032 * 4


def run_measured(n: int, args) -> dict:
    cmd = [sys.executable, "-m", "pytorch_distributed_training_amd.launch", f"--nproc_per_node={n}",
           "--master_addr=127.0.0.1", f"--master_port={29600 + n}", "--use_env", os.path.join(ROOT, "bench.py"),
           "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup)] + \
        (["--batch_size", str(args.batch_size)] if args.batch_size else []) + (["--sync_bn"] if args.sync_bn else [])
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    if out.returncode != 0:
        raise RuntimeError(f"N={n} failed:\n{out.stderr[-3000:]}")
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def model_curve(step_ms_1gpu: float, batch: int, sync_bn: bool, bucket_mb: float = 16.0, algo: str = "ring",
                syncbn_comm: str = "ipc"):
    """Predicted whole-job images/sec for N in {1,2,4,8}."""
    rows = []
    for n in (1, 2, 4, 8):
        if n == 1:
            t = step_ms_1gpu
        else:
            if algo == "ring":      # one link per direction: 2 (N-1)/N * S over one link
                bw_ms = 2.0 * (n - 1) / n * GRAD_BYTES / (LINK_GBPS * 1e9) * 1e3
            else:                   # two-shot direct over the full mesh: 2 S / N per link
                bw_ms = 2.0 * GRAD_BYTES / n / (LINK_GBPS * 1e9) * 1e3
            n_buckets = 1 + int((GRAD_BYTES / (1 << 20) + bucket_mb - 1) // bucket_mb)
            comm_ms = bw_ms + n_buckets * COLL_LATENCY_US * 1e-3
            backward_ms = step_ms_1gpu * 0.6  # what the all-reduce can hide under
            exposed = max(0.0, comm_ms - backward_ms) + bw_ms / n_buckets + COLL_LATENCY_US * 1e-3  # tail bucket
            per_msg_us = IPC_SMALL_US if syncbn_comm == "ipc" else COLL_LATENCY_US
            syncbn = 2 * 53 * per_msg_us * 1e-3 if sync_bn else 0.0
            t = step_ms_1gpu + exposed + syncbn
        rows.append({"n_gpus": n, "ms_per_step": round(t, 3), "images_per_sec": round(n * batch / t * 1e3, 1),
                     "efficiency": round(step_ms_1gpu / t, 4), "kind": "measured" if n == 1 else "modelled"})
    return rows


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch_size", type=int, default=None,
                    help="per-GPU batch; default: bench.py's own default (2560 for ResNet-50 at 224x224), so that the "
                         "curve is the headline benchmark's")
    ap.add_argument("--step_ms", type=float, default=None, help="model mode without a log: measured 1-GPU ms/step")
    ap.add_argument("--md", action="store_true", help="model mode: also print the curves as a markdown table")
    ap.add_argument("--sync_bn", action="store_true")
    ap.add_argument("--syncbn_comm", default="ipc", choices=["ipc", "rccl"],
                    help="modelled path of the SyncBN statistics all-reduce (parallel/ipc.py)")
    ap.add_argument("--model-from", dest="model_from", default=None, help="bench.py JSON line/log of a 1-GPU run")
    ap.add_argument("--calibration", default=None, help="JSON written by bench/micro/rccl_busbw.py --out")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    bucket_mb = 16.0
    if a.calibration:
        global LINK_GBPS, COLL_LATENCY_US
        cal = json.loads(open(a.calibration).read().strip().splitlines()[-1])
        LINK_GBPS, COLL_LATENCY_US = float(cal["link_GBps"]), float(cal["collective_latency_us"])
        bucket_mb = float(cal.get("bucket_mb", bucket_mb))
    if a.model_from or a.step_ms:
        if a.model_from:
            line = [l for l in open(a.model_from).read().splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            assert d["n_gpus"] == 1
            batch, step_ms = d["config"]["per_gpu_batch"], d["ms_per_step"]
        else:
            batch, step_ms = a.batch_size or 2560, a.step_ms
        result = {"source": a.model_from or f"--step_ms {a.step_ms}", "per_gpu_batch": batch, "sync_bn": a.sync_bn,
                  "link_GBps": LINK_GBPS, "collective_latency_us": COLL_LATENCY_US,
                  "constants": "calibrated: " + a.calibration if a.calibration else "task-brief defaults (unmeasured)",
                  "ring": model_curve(step_ms, batch, a.sync_bn, bucket_mb, "ring", a.syncbn_comm),
                  "direct": model_curve(step_ms, batch, a.sync_bn, bucket_mb, "direct", a.syncbn_comm)}
        if a.md:
            print(f"MODELLED (not measured): B={batch}/GPU, sync_bn={a.sync_bn}, {result['constants']}: link "
                  f"{LINK_GBPS:g} GB/s, {COLL_LATENCY_US:g} us per collective")
            print("| GPUs | ring all-reduce: img/s | efficiency | two-shot direct: img/s | efficiency | kind |")
            print("|---|---|---|---|---|---|")
            for r, dr in zip(result["ring"], result["direct"]):
                print(f"| {r['n_gpus']} | {r['images_per_sec']:.0f} | {r['efficiency']:.3f} | {dr['images_per_sec']:.0f} | "
                      f"{dr['efficiency']:.3f} | {r['kind']} |")
    else:
        # (no HIP call in this parent: it forks the launchers; see launch._gpu_count)
        sys.path.insert(0, ROOT)
        from pytorch_distributed_training_amd.launch import _gpu_count

        avail = _gpu_count()
        rows, base = [], None
        for n in a.gpus:
            if n > avail:
                print(f"skipping N={n}: only {avail} GPU(s) visible", file=sys.stderr)
                continue
            d = run_measured(n, a)
            base = base or d["value"] / d["n_gpus"]
            rows.append({"n_gpus": n, "images_per_sec": d["value"], "ms_per_step": d["ms_per_step"],
                         "efficiency": round(d["value"] / (n * base), 4), "kind": "measured"})
        result = {"measured": rows}
    text = json.dumps(result, indent=2)
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
