"""Process launcher: ``python -m pytorch_distributed_training_amd.launch``.

Same flag names and worker-environment contract as the launcher the reference
README uses (``python -m torch.distributed.launch --nproc_per_node=N
[--nnode(s)=M --node_rank=r --master_addr=A --master_port=P] main.py ...``,
``/root/reference/README.md:14,28,34``; contract in SURVEY §2.2 C1), written
from scratch as a small static-rendezvous process manager: one agent per node,
one worker per GPU, fail-fast group teardown, signal forwarding, optional
group restarts.

Worker env: ``LOCAL_RANK RANK GROUP_RANK ROLE_RANK ROLE_NAME LOCAL_WORLD_SIZE
WORLD_SIZE GROUP_WORLD_SIZE ROLE_WORLD_SIZE MASTER_ADDR MASTER_PORT
TORCHELASTIC_RESTART_COUNT TORCHELASTIC_MAX_RESTARTS TORCHELASTIC_RUN_ID``
with ``RANK = node_rank * nproc_per_node + local_rank``.
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import subprocess
import sys
import time
from dataclasses import dataclass, field
from typing import Sequence

_FORWARDED = (signal.SIGTERM, signal.SIGINT, signal.SIGHUP, signal.SIGQUIT)


def _parse_nnodes(value: str) -> int:
    """``--nnodes`` accepts ``N`` or ``MIN:MAX``; only a fixed size is supported."""
    parts = str(value).split(":")
    if len(parts) == 1:
        n = int(parts[0])
    elif len(parts) == 2:
        lo, hi = int(parts[0]), int(parts[1])
        if lo != hi:
            raise ValueError("elastic node ranges are not supported: use --nnodes=N (static rendezvous)")
        n = lo
    else:
        raise ValueError(f"bad --nnodes value {value!r}")
    if n < 1:
        raise ValueError("--nnodes must be >= 1")
    return n


def _gpu_count() -> int:
    """Usable GPUs WITHOUT touching the HIP runtime in this process: the launcher forks + execs every worker, and
    an exec out of a process that has initialised the GPU is refused on some pools (and leaks /dev/kfd state into
    the children).  Order: the visibility variables; the KFD topology in sysfs, counting only the nodes whose DRM
    render device this process can open (sysfs lists every GPU of the HOST, a container or cgroup may expose only
    some of them: their /dev/dri/renderD* nodes are then absent or not accessible); a short-lived child process
    asking ``torch.cuda.device_count()`` when sysfs does not say (no render minor recorded)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n = _gpu_count_sysfs()
    if n is not None:
        return n
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def _gpu_count_sysfs(topo: str = "/sys/class/kfd/kfd/topology/nodes", dri: str = "/dev/dri"):
    """GPU nodes of the KFD topology whose render device is accessible; None when the topology cannot answer
    (not readable, or a GPU node without a ``drm_render_minor``)."""
    try:
        nodes = os.listdir(topo)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(topo, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:  # CPU nodes report 0 SIMDs
                continue
            minor = int(props.get("drm_render_minor", "-1"))
        except (OSError, ValueError):
            continue
        if minor < 0:
            return None
        if os.access(os.path.join(dri, f"renderD{minor}"), os.R_OK | os.W_OK):
            n += 1
    return n


def determine_local_world_size(nproc_per_node: str) -> int:
    """int, or ``gpu`` / ``cpu`` / ``auto``."""
    try:
        n = int(nproc_per_node)
    except ValueError:
        key = nproc_per_node.lower()
        if key == "cpu":
            n = os.cpu_count() or 1
        elif key == "gpu":
            n = _gpu_count()
            if n == 0:
                raise ValueError("--nproc_per_node=gpu but no GPU is visible")
        elif key == "auto":
            n = _gpu_count() or (os.cpu_count() or 1)
        else:
            raise ValueError(f"unsupported --nproc_per_node {nproc_per_node!r}")
    if n < 1:
        raise ValueError("--nproc_per_node must be >= 1")
    return n


def get_args_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(
        prog="python -m pytorch_distributed_training_amd.launch",
        description="MI355X data-parallel launcher (one worker per GPU, static rendezvous)",
        allow_abbrev=True,  # `--nnode=2` in README.md:28,34 relies on prefix matching
    )
    p.add_argument("--nnodes", type=str, default="1:1", help="number of nodes (N or N:N)")
    p.add_argument("--nproc-per-node", "--nproc_per_node", dest="nproc_per_node", type=str, default="1",
                   help="workers per node: int, gpu, cpu or auto")
    p.add_argument("--node-rank", "--node_rank", dest="node_rank", type=int, default=0)
    p.add_argument("--master-addr", "--master_addr", dest="master_addr", type=str, default="127.0.0.1")
    p.add_argument("--master-port", "--master_port", dest="master_port", type=int, default=29500)
    p.add_argument("--max-restarts", "--max_restarts", dest="max_restarts", type=int, default=0)
    p.add_argument("--monitor-interval", "--monitor_interval", dest="monitor_interval", type=float, default=0.1)
    p.add_argument("--use-env", "--use_env", dest="use_env", action="store_true",
                   help="do not pass --local_rank; workers read LOCAL_RANK")
    p.add_argument("--local-rank-style", "--local_rank_style", dest="local_rank_style",
                   choices=("underscore", "hyphen"), default="underscore",
                   help="spell the injected flag --local_rank (reference main.py:24) or --local-rank (torch>=2)")
    p.add_argument("-m", "--module", action="store_true", help="treat training_script as a python module")
    p.add_argument("--no-python", "--no_python", dest="no_python", action="store_true",
                   help="exec training_script directly (requires --use-env)")
    p.add_argument("--standalone", action="store_true", help="single node on 127.0.0.1 with a free port")
    p.add_argument("--run-id", "--run_id", "--rdzv-id", "--rdzv_id", dest="run_id", type=str, default="none")
    p.add_argument("--log-dir", "--log_dir", dest="log_dir", type=str, default=None,
                   help="when set, each worker's stdout/stderr goes to <log_dir>/rank_<r>.{out,err}")
    p.add_argument("training_script", type=str)
    p.add_argument("training_script_args", nargs=argparse.REMAINDER)
    return p


def parse_args(argv: Sequence[str] | None = None) -> argparse.Namespace:
    args = get_args_parser().parse_args(argv)
    if args.no_python and not args.use_env:
        raise ValueError("--no-python requires --use-env (the flag cannot be injected into a non-python program)")
    return args


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@dataclass
class LaunchConfig:
    nnodes: int
    nproc_per_node: int
    node_rank: int
    master_addr: str
    master_port: int
    max_restarts: int = 0
    monitor_interval: float = 0.1
    run_id: str = "none"
    log_dir: str | None = None
    extra_env: dict = field(default_factory=dict)

    @property
    def world_size(self) -> int:
        return self.nnodes * self.nproc_per_node


def config_from_args(args: argparse.Namespace) -> LaunchConfig:
    nnodes = _parse_nnodes(args.nnodes)
    nproc = determine_local_world_size(args.nproc_per_node)
    if not 0 <= args.node_rank < nnodes:
        raise ValueError(f"--node_rank {args.node_rank} outside [0, {nnodes})")
    addr, port = args.master_addr, args.master_port
    if args.standalone:
        if nnodes != 1:
            raise ValueError("--standalone implies --nnodes=1")
        addr, port = "127.0.0.1", _free_port()
    return LaunchConfig(nnodes, nproc, args.node_rank, addr, port, args.max_restarts,
                        args.monitor_interval, args.run_id, args.log_dir)


def build_worker_cmd(args: argparse.Namespace, local_rank: int) -> list[str]:
    """``python -u script --local_rank=<r> <args>`` unless ``--use-env``."""
    cmd: list[str] = []
    if not args.no_python:
        cmd += [sys.executable, "-u"]
        if args.module:
            cmd.append("-m")
    cmd.append(args.training_script)
    if not args.use_env:
        flag = "--local_rank" if args.local_rank_style == "underscore" else "--local-rank"
        cmd.append(f"{flag}={local_rank}")
    cmd += list(args.training_script_args)
    return cmd


def build_worker_env(cfg: LaunchConfig, local_rank: int, restart_count: int, base: dict | None = None) -> dict:
    env = dict(os.environ if base is None else base)
    rank = cfg.node_rank * cfg.nproc_per_node + local_rank
    env.update({
        "LOCAL_RANK": str(local_rank),
        "RANK": str(rank),
        "GROUP_RANK": str(cfg.node_rank),
        "ROLE_RANK": str(rank),
        "ROLE_NAME": "default",
        "LOCAL_WORLD_SIZE": str(cfg.nproc_per_node),
        "WORLD_SIZE": str(cfg.world_size),
        "GROUP_WORLD_SIZE": str(cfg.nnodes),
        "ROLE_WORLD_SIZE": str(cfg.world_size),
        "MASTER_ADDR": cfg.master_addr,
        "MASTER_PORT": str(cfg.master_port),
        "TORCHELASTIC_RESTART_COUNT": str(restart_count),
        "TORCHELASTIC_MAX_RESTARTS": str(cfg.max_restarts),
        "TORCHELASTIC_RUN_ID": cfg.run_id,
    })
    # One thread per worker unless the user chose otherwise (avoids N x cores oversubscription).
    if "OMP_NUM_THREADS" not in env and cfg.nproc_per_node > 1:
        env["OMP_NUM_THREADS"] = "1"
    # dmabuf IPC is the only mode the MI355X host driver supports for RCCL peer mappings.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(cfg.extra_env)
    return env


class WorkerGroup:
    """The local workers of one node: spawn, poll, tear down."""

    def __init__(self, cfg: LaunchConfig, args: argparse.Namespace):
        self.cfg, self.args = cfg, args
        self.procs: list[subprocess.Popen] = []
        self._files = []

    def start(self, restart_count: int) -> None:
        self.procs = []
        for lr in range(self.cfg.nproc_per_node):
            env = build_worker_env(self.cfg, lr, restart_count)
            cmd = build_worker_cmd(self.args, lr)
            stdout = stderr = None
            if self.cfg.log_dir:
                os.makedirs(self.cfg.log_dir, exist_ok=True)
                rank = int(env["RANK"])
                stdout = open(os.path.join(self.cfg.log_dir, f"rank_{rank}.out"), "ab")
                stderr = open(os.path.join(self.cfg.log_dir, f"rank_{rank}.err"), "ab")
                self._files += [stdout, stderr]
            # own session => own process group: a kill reaches grandchildren too (no orphans)
            self.procs.append(subprocess.Popen(cmd, env=env, stdout=stdout, stderr=stderr, start_new_session=True))

    def poll(self) -> tuple[bool, int | None, int | None]:
        """(all_done, first_failed_local_rank, its_exit_code)."""
        all_done = True
        for lr, p in enumerate(self.procs):
            rc = p.poll()
            if rc is None:
                all_done = False
            elif rc != 0:
                return False, lr, rc
        return all_done, None, None

    def signal_all(self, sig: int) -> None:
        for p in self.procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def stop(self, grace_s: float = 5.0) -> None:
        self.signal_all(signal.SIGTERM)
        deadline = time.time() + grace_s
        for p in self.procs:
            try:
                p.wait(timeout=max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                pass
        self.signal_all(signal.SIGKILL)
        for p in self.procs:
            try:
                p.wait(timeout=5.0)
            except subprocess.TimeoutExpired:  # pragma: no cover
                pass
        for f in self._files:
            try:
                f.close()
            except Exception:
                pass
        self._files = []


def _exit_code(rc: int) -> int:
    # a worker killed by signal N reports -N; surface it the way a shell would
    return 128 - rc if rc < 0 else rc


def run(args: argparse.Namespace) -> int:
    cfg = config_from_args(args)
    group = WorkerGroup(cfg, args)
    received: list[int] = []

    def _handler(signum, _frame):
        received.append(signum)
        group.signal_all(signum)

    old = {}
    for s in _FORWARDED:
        try:
            old[s] = signal.signal(s, _handler)
        except (ValueError, OSError):  # not in main thread
            pass

    restarts = 0
    try:
        group.start(restarts)
        while True:
            done, failed_lr, rc = group.poll()
            if done:
                return 0
            if received:
                group.stop()
                return 128 + received[0]
            if failed_lr is not None:
                grank = cfg.node_rank * cfg.nproc_per_node + failed_lr
                print(f"[launch] worker rank {grank} (local {failed_lr}) failed with exit code {rc}; "
                      f"stopping the local group", file=sys.stderr, flush=True)
                group.stop()
                if restarts < cfg.max_restarts:
                    restarts += 1
                    print(f"[launch] restarting the local group ({restarts}/{cfg.max_restarts})",
                          file=sys.stderr, flush=True)
                    group.start(restarts)
                    continue
                return _exit_code(rc) or 1
            time.sleep(cfg.monitor_interval)
    finally:
        group.stop(grace_s=0.5) if any(p.poll() is None for p in group.procs) else None
        for s, h in old.items():
            signal.signal(s, h)


def main(argv: Sequence[str] | None = None) -> None:
    sys.exit(run(parse_args(argv)))


if __name__ == "__main__":
    main()
