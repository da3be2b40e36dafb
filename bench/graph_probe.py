"""Which part of the training step survives hipGraph capture?  Each stage runs in its own process
(a failing capture can take the process down) and prints one line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = ["ddp_fwd_only", "ddp_nosync", "ddp_aliased", "ddp_fwd_bwd", "ddp_py_reducer", "full_sgd"]


def stage(name):
    import torch
    sys.path.insert(0, ROOT)
    from pytorch_distributed_training_amd import dist as pdist
    from pytorch_distributed_training_amd import nn as pnn
    from pytorch_distributed_training_amd.engine import Engine, EngineConfig
    from pytorch_distributed_training_amd.ops import native

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    pdist.init_process_group(backend="rccl", local_rank=0)
    torch.manual_seed(0)
    x32 = torch.rand(32, 3, 32, 32, device="cuda")
    y = torch.randint(0, 10, (32,), device="cuda")

    def capture(fn, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        g.replay()
        torch.cuda.synchronize()
        return out

    if name == "cast":
        capture(lambda: native.nchw_to_nhwc_lowp(x32))
    elif name == "conv_fwd":
        x = torch.randn(8, 64, 16, 16, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.randn(64, 64, 3, 3, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        capture(lambda: native.conv2d_fwd(x, w, 1, 1, True))
    elif name == "bn_fwd":
        x = torch.randn(8, 64, 16, 16, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        g_, b_ = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")

        def f():
            m, i = native.bn_finalize(native.bn_stats(x), 8 * 256.0, 1e-5)
            return native.bn_apply_act(x, m, i, g_, b_, None, True)
        capture(f)
    elif name == "loss":
        lg = torch.randn(32, 10, device="cuda").bfloat16()
        capture(lambda: native.softmax_ce_fwd_bwd(lg, y))
    else:
        eng = Engine(EngineConfig(model="resnet18", num_classes=10, dtype="bf16",
                                  optimizer="adam" if name == "adam" else "sgd", lr=0.01, seed=0, sync_bn=False,
                                  reducer="py" if name == "ddp_py_reducer" else None))
        if name == "adam":
            eng.optimizer.set_capturable(True)
        xin = eng.prepare_input(x32)
        if name == "model_fwd":
            with torch.no_grad():
                capture(lambda: eng.net.module(xin))
        elif name == "model_fwd_bwd_nodpp":
            def f():
                for p in eng.net.module.parameters():
                    p.grad = None
                loss = eng.criterion(eng.net.module(xin), y)
                loss.backward()
                return loss
            eng.net.reducer.remove_hooks()
            capture(f)
        elif name in ("sgd", "adam"):
            for _ in range(3):
                eng.train_step(x32, y)
            capture(lambda: eng.optimizer.step(), warm=1)
        elif name == "ddp_fwd_only":
            capture(lambda: eng.criterion(eng.net(xin), y))
        elif name == "ddp_nosync":
            def f():
                with eng.net.no_sync():
                    loss = eng.criterion(eng.net(xin), y)
                    eng.optimizer.zero_grad()
                    loss.backward()
                return loss
            capture(f)
        elif name == "ddp_aliased":
            def f():  # grads stay the bucket views: the hooks find nothing to flatten
                loss = eng.criterion(eng.net(xin), y)
                loss.backward()
                return loss
            capture(f)
        elif name in ("ddp_fwd_bwd", "ddp_py_reducer"):
            def f():
                loss = eng.criterion(eng.net(xin), y)
                eng.optimizer.zero_grad()
                loss.backward()
                return loss
            capture(f)
        elif name == "full_sgd":
            capture(lambda: eng._eager_step(x32, y))
    print(f"STAGE {name}: OK", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        import faulthandler
        faulthandler.enable()
        stage(sys.argv[1])
    else:
        for s in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=240)
            tail = (r.stderr.strip().splitlines() or [""])
            key = [l for l in tail if "Error" in l or "error" in l or "Fatal" in l][:2]
            print(f"{s}: rc={r.returncode} {r.stdout.strip()[-60:]} {key}", flush=True)
