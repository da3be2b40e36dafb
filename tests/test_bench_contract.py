"""bench.py under the round driver's command line (CPU rehearsal with gloo, world size 1 and 2).

The driver launches ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N --steps K --warmup W`` and reads ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config"}
SMALL = ["--backend", "gloo", "--model", "resnet18", "--batch_size", "2", "--image_size", "32", "--num_classes", "10",
         "--dtype", "fp32", "--steps", "2", "--warmup", "1"]


def _json_lines(out: str):
    rows = []
    for line in out.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            rows.append(json.loads(line))
    return rows


def _env():
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["PDT_FORCE_REFERENCE_OPS"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_single_process_prints_one_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *SMALL], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = _json_lines(p.stdout)
    assert len(rows) == 1
    r = rows[0]
    assert KEYS <= set(r) and r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["data"] == "synthetic"
    assert r["config"]["global_batch"] == 2 and r["config"]["parallelism"] == "dp1"
    assert abs(r["value"] - r["config"]["global_batch"] / (r["ms_per_step"] / 1e3)) / r["value"] < 1e-2


@pytest.mark.timeout(600)
def test_two_ranks_under_torch_distributed_run(free_port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL]
    p = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=540)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = _json_lines(p.stdout)
    assert len(rows) == 1, p.stdout[-2000:]  # rank 0 only
    r = rows[0]
    assert KEYS <= set(r) and r["n_gpus"] == 2
    assert r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2"  # value is the whole job's
    assert abs(r["value"] - 4 / (r["ms_per_step"] / 1e3)) / r["value"] < 1e-2


def test_gpus_flag_must_match_world_size(free_port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port), os.path.join(ROOT, "bench.py"), "--gpus", "4", *SMALL]
    p = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def test_rccl_busbw_calibration_feeds_the_scaling_model(tmp_path, free_port):
    """bench/micro/rccl_busbw.py (all-reduce sweep -> link_GBps / latency / bucket size) on 2 gloo ranks, and
    bench/scaling.py --calibration replacing its task-brief constants with that file."""
    cal = tmp_path / "cal.json"
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench", "micro", "rccl_busbw.py"), "--gpus", "2",
                          "--backend", "gloo", "--device", "cpu", "--sizes_mb", "1", "2", "--iters", "3", "--warmup", "1",
                          "--port", str(free_port), "--out", str(cal)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    doc = json.loads(cal.read_text())
    assert doc["world"] == 2 and doc["link_GBps"] > 0 and doc["collective_latency_us"] > 0 and doc["bucket_mb"] in (1, 2)
    assert len(doc["rows"]) == 1 and len(doc["rows"][0]["rows"]) == 3  # 4 KiB latency row + two sizes
    line = tmp_path / "bench.log"
    line.write_text(json.dumps({"n_gpus": 1, "ms_per_step": 95.0, "config": {"per_gpu_batch": 1280}}) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench", "scaling.py"), "--model-from", str(line),
                          "--calibration", str(cal)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout)
    assert res["link_GBps"] == doc["link_GBps"] and res["constants"].startswith("calibrated")
    assert [r["n_gpus"] for r in res["ring"]] == [1, 2, 4, 8]


def test_convergence_script_runs_on_cpu(tmp_path):
    """bench/convergence.py (native vs stock vs oracle learning curves on the GPU) stays runnable: on CPU every back
    end is the fp32 oracle, so the rows must agree exactly."""
    import json

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench", "convergence.py"), "--model", "resnet18", "--steps", "4",
                        "--window", "2", "--eval_batches", "1", "--batch_size", "8", "--num_classes", "10",
                        "--backends", "native,oracle"], cwd=tmp_path, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert [r["backend"] for r in rows] == ["native", "oracle"] and all(r["finite"] for r in rows)
    assert rows[0]["loss_curve"] == rows[1]["loss_curve"] and "held-out top-1" in p.stdout


def test_cli_matrix_verdict_line(tmp_path):
    (tmp_path / "s.json").write_text('{"images_per_sec_synchronised": 123.456, "hip_graph": true, "reducer": "native"}')
    (tmp_path / "s.log").write_text("Epoch: 0 step: 0 loss: 4.5\nEpoch: 0 step: 10 loss: 0.5\n")
    v = os.path.join(ROOT, "bench", "cli_matrix_verdict.py")
    p = subprocess.run([sys.executable, v, str(tmp_path / "s.json"), str(tmp_path / "s.log")], capture_output=True, text=True)
    assert p.stdout.startswith("OK loss 4.5 -> 0.5") and '"hip_graph": true' in p.stdout
    (tmp_path / "s.log").write_text("Epoch: 0 step: 0 loss: 4.5\nEpoch: 0 step: 10 loss: nan\n")
    p = subprocess.run([sys.executable, v, str(tmp_path / "s.json"), str(tmp_path / "s.log")], capture_output=True, text=True)
    assert p.stdout.startswith("CHECK")
    p = subprocess.run([sys.executable, v, str(tmp_path / "missing.json"), str(tmp_path / "s.log"), "finite-only"],
                       capture_output=True, text=True)
    assert p.stdout.startswith("CHECK") and "missing" in p.stdout
