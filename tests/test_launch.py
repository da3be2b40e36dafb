"""T0/T2: launcher argument parsing, env contract, multi-"node" ranks, fail-fast."""
import os
import subprocess
import sys
import textwrap
import time

import pytest

from pytorch_distributed_training_amd import launch


def test_defaults_match_reference_launcher():
    a = launch.parse_args(["main.py"])
    cfg = launch.config_from_args(a)
    assert (cfg.nnodes, cfg.nproc_per_node, cfg.node_rank) == (1, 1, 0)
    assert (cfg.master_addr, cfg.master_port, cfg.max_restarts) == ("127.0.0.1", 29500, 0)
    assert not a.use_env and not a.module and not a.no_python


def test_readme_single_node_recipe():
    # README.md:14  --nproc_per_node=ngpus --master_port=29500 main.py ...
    a = launch.parse_args(["--nproc_per_node=4", "--master_port=29500", "main.py", "--batch_size", "64"])
    cfg = launch.config_from_args(a)
    assert cfg.nproc_per_node == 4 and cfg.world_size == 4
    assert a.training_script == "main.py" and a.training_script_args == ["--batch_size", "64"]


def test_readme_two_node_recipe_with_nnode_typo():
    # README.md:28,34 spells --nnode=2 (prefix of --nnodes) and uses underscores
    a = launch.parse_args(["--nproc_per_node=4", "--nnode=2", "--node_rank=1", "--master_addr=10.0.0.1",
                           "--master_port=29500", "main.py"])
    cfg = launch.config_from_args(a)
    assert cfg.nnodes == 2 and cfg.node_rank == 1 and cfg.world_size == 8
    env = launch.build_worker_env(cfg, local_rank=3, restart_count=0, base={})
    assert env["RANK"] == "7" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8"
    assert env["GROUP_RANK"] == "1" and env["LOCAL_WORLD_SIZE"] == "4" and env["GROUP_WORLD_SIZE"] == "2"
    assert env["MASTER_ADDR"] == "10.0.0.1" and env["MASTER_PORT"] == "29500"


def test_hyphen_and_underscore_flags_are_equivalent():
    a = launch.parse_args(["--nproc-per-node=2", "--node-rank=0", "--master-addr=h", "--master-port=1", "x.py"])
    b = launch.parse_args(["--nproc_per_node=2", "--node_rank=0", "--master_addr=h", "--master_port=1", "x.py"])
    assert vars(a) == vars(b)


def test_env_contract_is_complete():
    cfg = launch.LaunchConfig(1, 2, 0, "127.0.0.1", 1234, max_restarts=3, run_id="abc")
    env = launch.build_worker_env(cfg, 1, 2, base={})
    for k in ("LOCAL_RANK", "RANK", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE", "WORLD_SIZE",
              "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RESTART_COUNT",
              "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_RUN_ID"):
        assert k in env, k
    assert env["TORCHELASTIC_RESTART_COUNT"] == "2" and env["TORCHELASTIC_MAX_RESTARTS"] == "3"
    assert env["OMP_NUM_THREADS"] == "1"  # nproc > 1 and unset
    assert launch.build_worker_env(launch.LaunchConfig(1, 1, 0, "h", 1), 0, 0, base={}).get("OMP_NUM_THREADS") is None
    assert launch.build_worker_env(cfg, 0, 0, base={"OMP_NUM_THREADS": "7"})["OMP_NUM_THREADS"] == "7"


def test_worker_command_forms():
    a = launch.parse_args(["main.py", "--x", "1"])
    assert launch.build_worker_cmd(a, 2) == [sys.executable, "-u", "main.py", "--local_rank=2", "--x", "1"]
    a = launch.parse_args(["--local_rank_style", "hyphen", "main.py"])
    assert launch.build_worker_cmd(a, 0)[-1] == "--local-rank=0"
    a = launch.parse_args(["--use_env", "main.py"])
    assert launch.build_worker_cmd(a, 5) == [sys.executable, "-u", "main.py"]
    a = launch.parse_args(["-m", "pkg.mod", "--flag"])
    assert launch.build_worker_cmd(a, 1) == [sys.executable, "-u", "-m", "pkg.mod", "--local_rank=1", "--flag"]
    a = launch.parse_args(["--no_python", "--use_env", "./prog", "a"])
    assert launch.build_worker_cmd(a, 0) == ["./prog", "a"]
    with pytest.raises(ValueError):
        launch.parse_args(["--no_python", "./prog"])


def test_nnodes_and_nproc_parsing():
    assert launch._parse_nnodes("3") == 3 and launch._parse_nnodes("2:2") == 2
    with pytest.raises(ValueError):
        launch._parse_nnodes("1:4")
    assert launch.determine_local_world_size("3") == 3
    assert launch.determine_local_world_size("cpu") == (os.cpu_count() or 1)
    assert launch.determine_local_world_size("auto") >= 1
    with pytest.raises(ValueError):
        launch.determine_local_world_size("tpu")


def _run_launcher(args, script_body, tmp_path, timeout=60):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(script_body))
    cmd = [sys.executable, "-m", "pytorch_distributed_training_amd.launch", *args, str(script)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=tmp_path)


def test_spawns_workers_with_ranks(tmp_path, free_port):
    r = _run_launcher(["--nproc_per_node=3", f"--master_port={free_port}"], """
        import argparse, os
        p = argparse.ArgumentParser(); p.add_argument('--local_rank', type=int); a = p.parse_args()
        assert a.local_rank == int(os.environ['LOCAL_RANK'])
        open(f"out_{os.environ['RANK']}.txt", 'w').write(os.environ['WORLD_SIZE'] + ' ' + os.environ['MASTER_PORT'])
    """, tmp_path)
    assert r.returncode == 0, r.stderr
    for rank in range(3):
        assert (tmp_path / f"out_{rank}.txt").read_text() == f"3 {free_port}"


def test_two_nodes_on_localhost_rendezvous(tmp_path, free_port):
    """T2b: README.md:28,34 as two launcher invocations on one box; global ranks + store rendezvous."""
    body = """
        import os, torch, torch.distributed as dist
        dist.init_process_group('gloo', init_method='env://')
        t = torch.tensor([float(dist.get_rank())])
        dist.all_reduce(t)
        open(f"sum_{dist.get_rank()}.txt", 'w').write(f"{t.item()} {dist.get_world_size()} {os.environ['GROUP_RANK']}")
        dist.destroy_process_group()
    """
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(body))
    procs = []
    for node in range(2):
        cmd = [sys.executable, "-m", "pytorch_distributed_training_amd.launch", "--nproc_per_node=2", "--nnode=2",
               f"--node_rank={node}", "--master_addr=127.0.0.1", f"--master_port={free_port}", "--use_env", str(script)]
        procs.append(subprocess.Popen(cmd, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err
    for rank in range(4):
        s, w, g = (tmp_path / f"sum_{rank}.txt").read_text().split()
        assert float(s) == 6.0 and w == "4" and g == str(rank // 2)


def test_worker_failure_tears_group_down(tmp_path, free_port):
    """Fault injection: one worker dies mid-run -> non-zero exit quickly, no orphans."""
    t0 = time.time()
    r = _run_launcher(["--nproc_per_node=2", f"--master_port={free_port}", "--use_env"], """
        import os, sys, time
        open(f"pid_{os.environ['RANK']}.txt", 'w').write(str(os.getpid()))
        if os.environ['RANK'] == '1':
            time.sleep(0.5); sys.exit(7)
        time.sleep(120)
    """, tmp_path, timeout=60)
    assert r.returncode == 7
    assert time.time() - t0 < 30
    assert "failed with exit code 7" in r.stderr
    pid0 = int((tmp_path / "pid_0.txt").read_text())
    time.sleep(0.2)
    with pytest.raises(ProcessLookupError):
        os.kill(pid0, 0)


def test_max_restarts_restarts_the_group(tmp_path, free_port):
    r = _run_launcher(["--nproc_per_node=2", f"--master_port={free_port}", "--use_env", "--max_restarts=1"], """
        import os, sys
        n = int(os.environ['TORCHELASTIC_RESTART_COUNT'])
        open(f"try_{os.environ['RANK']}_{n}.txt", 'w').write('x')
        if n == 0 and os.environ['RANK'] == '0':
            sys.exit(3)
    """, tmp_path)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "try_0_1.txt").exists() and (tmp_path / "try_1_1.txt").exists()
