"""Where does the host time of the eager (no hipGraph) CIFAR-shaped step go?  cProfile over N steps of the reference
script's configuration (ResNet-50, 3x32x32, B=128, Adam, SyncBN modules) + wall time per step with the GPU idle-waiting."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from pytorch_distributed_training_amd import dist as pdist
    from pytorch_distributed_training_amd.engine import Engine, EngineConfig

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29588")
    pdist.init_process_group(backend="rccl", local_rank=0)
    eng = Engine(EngineConfig(model="resnet50", num_classes=1000, dtype="bf16", optimizer="adam", lr=1e-3, sync_bn=True,
                              seed=0, image_size=32, graph=False))
    x = torch.rand(128, 3, 32, 32).cuda()
    y = torch.randint(0, 100, (128,)).cuda()
    for _ in range(10):
        eng.train_step(x, y)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        eng.train_step(x, y)
    t_host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e3
    print(f"eager step: host {t_host:.3f} ms, synchronised {t_all:.3f} ms", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        eng.train_step(x, y)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(35)
    st.sort_stats("cumulative").print_stats(45)
    eng.close()
    pdist.shutdown()


if __name__ == "__main__":
    main()
