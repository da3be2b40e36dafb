"""T2: multi-process CPU/gloo tests of the reducers, SyncBN, buffer sync, reduce_loss, checkpoint
(SURVEY §4.2).  The same reducer logic runs on the GPU with RCCL collectives."""
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

import pytorch_distributed_training_amd as pdt
from pytorch_distributed_training_amd import nn as pnn
from pytorch_distributed_training_amd.parallel import initial_order, plan_buckets

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_workers as W  # noqa: E402

REDUCERS = ["py"] + (["native"] if pdt.native_available() else [])


def spawn(fn_name, world, port, tmp_path, **kwargs):
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=W.run, args=(fn_name, r, world, port, str(tmp_path), kwargs)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    errs = [open(tmp_path / f"error_{r}.txt").read() for r in range(world) if (tmp_path / f"error_{r}.txt").exists()]
    assert not errs, "\n".join(errs)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [torch.load(tmp_path / f"result_{r}.pt", weights_only=False) for r in range(world)]


def full_batch_oracle(sync_bn=True):
    """One process, whole batch, rank-0 init: what DDP(+SyncBN) must reproduce."""
    net = W._toy(seed=0)
    x, y = W._data()
    loss = pnn.CrossEntropyLoss()(net(x), y)
    loss.backward()
    return net, float(loss.detach())


# ------------------------------------------------------------------------- planning
def test_bucket_plan_sizes_and_order():
    numels = [100, 200, 300, 400, 50]
    order = initial_order(5)
    assert order == [4, 3, 2, 1, 0]  # reverse registration order
    b = plan_buckets(numels, order, 4, cap_bytes=3000, first_cap_bytes=600)
    # slots are padded to 128 elements: 512, 2048, 1536, 1024, 512 bytes in this order
    assert [s.param_indices for s in b] == [[4], [3], [2, 1], [0]]
    assert all(o % 128 == 0 for s in b for o in s.offsets)
    big = plan_buckets([10_000, 10], [0, 1], 4, cap_bytes=1000)
    assert [s.param_indices for s in big] == [[0], [1]]  # an oversize tensor gets its own bucket


def test_resnet50_bucket_count_is_xgmi_sized():
    from pytorch_distributed_training_amd.models import resnet50

    numels = [p.numel() for p in resnet50().parameters()]
    b = plan_buckets(numels, initial_order(len(numels)), 4, 16 << 20, 1 << 20)
    assert 6 <= len(b) <= 9  # ~102 MB of fp32 grads in 16 MiB buckets + small first bucket
    assert b[0].numel * 4 <= 9 << 20  # fc.weight (8 MB) rides alone / first
    assert sorted(i for s in b for i in s.param_indices) == list(range(len(numels)))


# --------------------------------------------------------------------- DDP gradients
@pytest.mark.parametrize("reducer", REDUCERS)
@pytest.mark.parametrize("world", [2, 4])
def test_ddp_syncbn_grads_equal_full_batch(reducer, world, free_port, tmp_path):
    res = spawn("ddp_grads", world, free_port, tmp_path, reducer=reducer, sync_bn=True, bucket_cap_mb=0.0005)
    oracle, oloss = full_batch_oracle()
    for r in res:
        assert r["kind"] == reducer
        assert len(r["buckets"]) > 1  # tiny cap: several buckets, exercised in order
        for n, p in oracle.named_parameters():
            assert torch.allclose(r["params"][n], p.detach(), atol=0), n       # ctor broadcast of rank 0's init
            assert torch.allclose(r["grads"][n], p.grad, atol=2e-5, rtol=1e-4), n
        for n, b in oracle.named_buffers():  # SyncBN running stats == full-batch BN running stats
            assert torch.allclose(r["buffers"][n].float(), b.float(), atol=1e-5), n
        assert r["grad_is_view"]  # .grad aliases the flat bucket
    # averaged loss over ranks equals the full-batch loss (equal shard sizes)
    assert abs(sum(r["loss"] for r in res) / world - oloss) < 1e-5


@pytest.mark.parametrize("reducer", REDUCERS)
def test_bf16_gradient_buckets(reducer, free_port, tmp_path):
    """``grad_dtype=torch.bfloat16``: the buckets (what crosses the links) are bf16, ``.grad`` stays fp32 and equal on
    every rank; the values match the full-batch gradients to bf16 precision."""
    res = spawn("ddp_grads", 2, free_port, tmp_path, reducer=reducer, sync_bn=True, bucket_cap_mb=0.0005, steps=2,
                grad_dtype="bfloat16")
    oracle, _ = full_batch_oracle()
    assert res[0]["bucket_dtype"] == "torch.bfloat16" and not res[0]["grad_is_view"]
    for n, p in oracle.named_parameters():
        g0, g1 = res[0]["grads"][n], res[1]["grads"][n]
        assert g0.dtype == torch.float32 and torch.equal(g0, g1), n
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((g0 - p.grad).abs().max()) / scale < 2e-2, n


@pytest.mark.parametrize("reducer", REDUCERS)
def test_bucket_rebuild_after_first_iteration(reducer, free_port, tmp_path):
    res = spawn("ddp_grads", 2, free_port, tmp_path, reducer=reducer, sync_bn=True, bucket_cap_mb=0.0005, steps=3)
    oracle, _ = full_batch_oracle()
    assert res[0]["rebuilt"] and res[1]["rebuilt"]
    assert res[0]["buckets"] == res[1]["buckets"]  # every rank derived the same plan (S6)
    order = [i for b in res[0]["buckets"] for i in b]
    assert order[0] in (6, 7)  # gradient-ready order: fc first
    for n, p in oracle.named_parameters():
        assert torch.allclose(res[0]["grads"][n], p.grad, atol=2e-5, rtol=1e-4), n


@pytest.mark.parametrize("reducer", REDUCERS)
def test_no_sync_accumulates_locally(reducer, free_port, tmp_path):
    res = spawn("ddp_grads", 2, free_port, tmp_path, reducer=reducer, sync_bn=False, steps=2, no_sync_first=True)
    a, b = res[0]["local_only"], res[1]["local_only"]
    assert any(not torch.allclose(x, y) for x, y in zip(a, b))  # inside no_sync nothing was averaged
    for n in res[0]["grads"]:
        assert torch.allclose(res[0]["grads"][n], res[1]["grads"][n], atol=1e-7), n  # synced step: identical


@pytest.mark.parametrize("reducer", REDUCERS)
@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_training_equals_single_process(reducer, optimizer, free_port, tmp_path):
    from pytorch_distributed_training_amd.optim import build_optimizer

    res = spawn("train_equivalence", 2, free_port, tmp_path, reducer=reducer, optimizer=optimizer)
    net = W._toy(seed=7)
    opt = build_optimizer(optimizer, net.parameters(), lr=0.05)
    crit = pnn.CrossEntropyLoss()
    for s in range(4):
        x, y = W._data(seed=100 + s)
        loss = crit(net(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
    for n, p in net.named_parameters():
        assert torch.allclose(res[0]["params"][n], p.detach(), atol=5e-5, rtol=1e-4), n
        assert torch.equal(res[0]["params"][n], res[1]["params"][n]), n  # replicas bit-identical
    for n, b in net.named_buffers():
        assert torch.allclose(res[0]["buffers"][n].float(), b.float(), atol=1e-5), n


def test_reduce_loss_semantics(free_port, tmp_path):
    """main.py:16-20: rank 0 ends with the mean; other ranks keep a local value."""
    res = spawn("reduce_loss_semantics", 3, free_port, tmp_path)
    assert res[0] == pytest.approx((1 + 2 + 3) / 3)


def test_buffer_broadcast_for_plain_bn(free_port, tmp_path):
    res = spawn("buffer_broadcast", 2, free_port, tmp_path)
    a, b = res[0]["buffers"], res[1]["buffers"]
    # after the second forward both ranks started from rank 0's buffers; the stats then absorbed one
    # more (different) local batch, so they differ by at most one momentum step
    assert int(a["bn1.num_batches_tracked"]) == 2 and int(b["bn1.num_batches_tracked"]) == 2
    assert not torch.equal(a["bn1.running_mean"], b["bn1.running_mean"])
    assert (a["bn1.running_mean"] - b["bn1.running_mean"]).abs().max() < 0.2


@pytest.mark.skipif(not pdt.native_available(), reason="extension not built")
def test_checkpoint_save_resume(free_port, tmp_path):
    path = str(tmp_path / "ckpt.pt")
    res = spawn("checkpoint_roundtrip", 2, free_port, tmp_path, path=path)
    assert [r["wrote"] for r in res] == [True, False]  # rank 0 only
    for r in res:
        assert r["meta"]["epoch"] == 1 and r["meta"]["global_step"] == 2
        assert r["got"] == pytest.approx(r["expect"], abs=1e-6)
        assert r["same"]
    ck = torch.load(path, weights_only=True)
    assert set(ck) >= {"model", "optimizer", "epoch", "global_step", "format_version"}
    assert all(not k.startswith("module.") for k in ck["model"])
    assert all(v.dtype in (torch.float32, torch.int64) and v.is_contiguous() for v in ck["model"].values())
    assert set(ck["optimizer"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    # loads with strict=True into a stock-structured replica
    fresh = W._toy(seed=1)
    fresh.load_state_dict(ck["model"], strict=True)


# ----------------------------------------------------------------- end-to-end script
def test_main_py_two_ranks_gloo(tmp_path, free_port, repo_root):
    """BASELINE.json config 1: ResNet-18, CIFAR-10-shaped synthetic data, DDP over gloo, W=2,
    through OUR launcher and main.py; checks the per-rank log files, checkpoint and resume."""
    cmd = [sys.executable, "-m", "pytorch_distributed_training_amd.launch", "--nproc_per_node=2",
           f"--master_port={free_port}", os.path.join(repo_root, "main.py"), "--model", "resnet18", "--dataset",
           "synthetic-cifar10", "--dataset_len", "640", "--num_classes", "10", "--batch_size", "32", "--epochs", "1",
           "--profile", "none", "--backend", "gloo", "--JobID", "T", "--checkpoint", "ck.pt", "--summary_json", "s.json",
           "--grad_comm", "auto"]  # (CPU tensors: the IPC bucket path declines, the group carries the buckets)
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("Start") == 2 and "Epoch: 0 step: 0 loss:" in r.stdout
    for rank in range(2):
        lines = open(tmp_path / f"T_32_{rank}.log").read().splitlines()
        assert lines[0] == "datetime\tg_step\tg_img\tloss_value\texamples_per_sec"
        assert lines[-1].startswith("TrainTime\t")
        rows = lines[1:-1]
        assert len(rows) == (2 if rank == 0 else 0)  # 10 steps per rank -> rows at steps 5 and 10, rank 0 only
    g = [l.split("\t") for l in open(tmp_path / "T_32_0.log").read().splitlines()[1:-1]]
    assert [x[1] for x in g] == ["10", "20"] and [x[2] for x in g] == ["320", "640"]
    assert float(g[1][3]) < 2.5
    import json

    s = json.load(open(tmp_path / "s.json"))
    assert s["world_size"] == 2 and s["steps"] == 10
    assert s["images_per_sec_synchronised"] > 0 and "comm" not in s  # (bucket timing needs the RCCL communicator)
    # resume for a second epoch
    r2 = subprocess.run(cmd[:-2] + ["--resume", "ck.pt", "--epochs", "2", "--eval"], cwd=tmp_path, capture_output=True,
                        text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert "Epoch: 1 step: 0" in r2.stdout and "Epoch: 0 step: 0" not in r2.stdout
    # the reference's commented-out evaluation loop (main.py:119-130), sharded over the ranks, printed by rank 0
    acc = [l for l in r2.stdout.splitlines() if l.startswith("eval accuracy:")]
    assert len(acc) == 1 and 0.0 <= float(acc[0].split(":")[1]) <= 1.0


def test_bucket_cap_autotune_agrees_across_ranks(free_port, tmp_path):
    res = spawn("bucket_autotune", 2, free_port, tmp_path)
    picks = [r["tune"]["picked_mb"] for r in res]
    assert picks[0] == picks[1] and picks[0] in (4.0, 8.0, 16.0, 32.0, 64.0)
    assert set(res[0]["tune"]["seconds_per_allreduce"]) == {"4", "8", "16", "32", "64"}
    assert res[0]["buckets"] == res[1]["buckets"]


@pytest.mark.parametrize("reducer", REDUCERS)
def test_reducer_debug_mode_and_replica_check(reducer, free_port, tmp_path):
    """the debug assertions run (and pass) every iteration, buckets go out in plan order on both ranks,
    and a parameter diverged on one rank is caught by the checksum all-gather on EVERY rank."""
    res = spawn("reducer_debug_mode", 2, free_port, tmp_path, reducer=reducer, diverge=True)
    for r in res:
        assert r["buckets"] >= 3
        assert all(log == list(range(len(log))) and len(log) >= 3 for log in r["logs"])
        assert r["checks"] >= 3 * r["buckets"]
        assert r["caught"] is not None and "replica divergence" in r["caught"]
    assert res[0]["logs"][-1] == res[1]["logs"][-1]


def test_ordered_shutdown_two_ranks(free_port, tmp_path):
    res = spawn("engine_shutdown", 2, free_port, tmp_path)
    assert all(not r["initialized_after"] for r in res)
    assert all(l == l for r in res for l in r["losses"])


def test_main_py_refuses_a_head_narrower_than_the_label_range_and_creates_log_dir(tmp_path, free_port, repo_root):
    """Two usability faults bench/cli_matrix.sh found: a classifier head narrower than the dataset's labels trained
    on garbage (now refused up front), and a ``--log_dir`` that did not exist yet crashed the throughput log."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PYTHONPATH=str(repo_root))
    base = [sys.executable, os.path.join(str(repo_root), "main.py"), "--backend", "gloo", "--model", "resnet18", "--batch_size", "8",
            "--dataset_len", "32", "--epochs", "1", "--steps", "2", "--profile", "none"]
    r = subprocess.run(base + ["--dataset", "synthetic-cifar", "--num_classes", "10"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "narrower than the 100 classes" in r.stderr, r.stderr[-2000:]
    r = subprocess.run(base + ["--dataset", "synthetic-cifar10", "--num_classes", "10", "--log_dir", "logs/a/b"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "logs" / "a" / "b" / "Job0_8_0.log").exists()


def test_main_py_mid_epoch_resume_matches_uninterrupted_run(tmp_path, free_port, repo_root):
    """--checkpoint_every saves a position INSIDE the epoch; --resume continues from it: same number of steps in
    total, same final parameters as the run that was never interrupted."""
    def run(extra, port, cwd):
        os.makedirs(cwd, exist_ok=True)
        cmd = [sys.executable, "-m", "pytorch_distributed_training_amd.launch", "--nproc_per_node=2",
               f"--master_port={port}", os.path.join(repo_root, "main.py"), "--backend", "gloo", "--model", "resnet18",
               "--dataset", "synthetic-cifar10", "--num_classes", "10", "--dataset_len", "64", "--batch_size", "4",
               "--dtype", "fp32", "--profile", "none", "--epochs", "2", "--seed", "0", "--optimizer", "sgd",
               "--no_graph"] + extra
        env = dict(os.environ, PYTHONPATH=repo_root, OMP_NUM_THREADS="1")
        r = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
        return r.stdout

    full, part, resumed = str(tmp_path / "full.pt"), str(tmp_path / "part.pt"), str(tmp_path / "resumed.pt")
    run(["--checkpoint", full], free_port, str(tmp_path / "a"))          # 2 epochs x 8 steps, uninterrupted
    run(["--checkpoint", part, "--checkpoint_every", "3", "--steps", "9"], free_port + 1, str(tmp_path / "b"))
    st9 = torch.load(part, weights_only=True)
    # step 9 = first batch of epoch 1: the checkpoint records the position inside the epoch
    assert st9["global_step"] == 9 and st9["epoch"] == 1 and st9["extra"]["samples_in_epoch"] == 4
    out = run(["--checkpoint", resumed, "--resume", part], free_port + 2, str(tmp_path / "c"))
    assert "Epoch: 1 step: 0 " not in out  # the batch that was already trained is not replayed
    a = torch.load(full, weights_only=True)
    b = torch.load(resumed, weights_only=True)
    assert a["global_step"] == b["global_step"] == 16 and a["epoch"] == b["epoch"] == 2
    for k, v in a["model"].items():
        assert torch.allclose(v.float(), b["model"][k].float(), atol=1e-5, rtol=1e-4), k
