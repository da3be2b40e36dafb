"""Per-rank TSV throughput log + machine-readable JSON summary.

Reference format (``/root/reference/main.py:65-67,107-111,117``):
file ``{JobID}_{batch_size}_{global_rank}.log`` in the CWD on every rank,
header ``datetime\\tg_step\\tg_img\\tloss_value\\texamples_per_sec``, rows from
rank 0 every 5 global steps with ``g_step = step*W``, ``g_img = step*W*B``,
``examples_per_sec = B/duration`` (per-GPU, last iteration), footer
``TrainTime\\t%f``.  Kept byte-compatible; the JSON summary adds the correctly
synchronised whole-job numbers.
"""
from __future__ import annotations

import json
import time
from datetime import datetime

HEADER = "datetime\tg_step\tg_img\tloss_value\texamples_per_sec\n"


def log_file_name(job_id: str, batch_size: int, global_rank: int) -> str:
    return f"{job_id}_{batch_size}_{global_rank}.log"


def format_row(now, global_step: int, world_size: int, batch_size: int, loss_value: float,
               examples_per_sec: float) -> str:
    return f"{now}\t{global_step * world_size}\t{global_step * world_size * batch_size}\t{loss_value}\t{examples_per_sec}\n"


class ThroughputLog:
    def __init__(self, job_id: str, batch_size: int, global_rank: int, world_size: int, directory: str = "."):
        import os

        os.makedirs(directory or ".", exist_ok=True)  # (every rank writes its own file; a missing --log_dir is created)
        self.path = os.path.join(directory, log_file_name(job_id, batch_size, global_rank))
        self.batch_size, self.rank, self.world = batch_size, global_rank, world_size
        self.f = open(self.path, "w")
        self.f.write(HEADER)
        self.f.flush()
        self.train_begin = time.time()

    def should_log(self, global_step: int) -> bool:
        return self.rank == 0 and global_step % 5 == 0

    def row(self, global_step: int, loss_value: float, duration_s: float) -> None:
        eps = self.batch_size / duration_s if duration_s > 0 else float("inf")
        self.f.write(format_row(datetime.now(), global_step, self.world, self.batch_size, loss_value, eps))

    def footer(self) -> None:
        self.f.write("TrainTime\t%f\n" % (time.time() - self.train_begin))
        self.f.flush()

    def close(self) -> None:
        if not self.f.closed:
            self.f.close()


class StepTimer:
    """Synchronised step-time statistics for the JSON summary."""

    def __init__(self):
        self.samples: list[float] = []

    def add(self, seconds: float) -> None:
        self.samples.append(seconds)

    def summary(self, images_per_step_global: int) -> dict:
        if not self.samples:
            return {}
        s = sorted(self.samples)
        n = len(s)
        mean = sum(s) / n
        return {
            "steps": n,
            "step_ms_mean": mean * 1e3,
            "step_ms_p50": s[n // 2] * 1e3,
            "step_ms_p90": s[min(n - 1, int(n * 0.9))] * 1e3,
            "step_ms_max": s[-1] * 1e3,
            "images_per_sec": images_per_step_global / mean,
        }


def write_json(path: str, obj: dict) -> None:
    with open(path, "w") as f:
        json.dump(obj, f, indent=2, sort_keys=True)
        f.write("\n")
