"""Where does the host time of an eager step go on the launch-bound CIFAR-shaped config (ResNet-50, B=128, 32x32)?"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.engine import Engine, EngineConfig  # noqa: E402

eng = Engine(EngineConfig(model="resnet50", num_classes=1000, optimizer="adam", lr=1e-3, sync_bn=True, seed=0),
             device=torch.device("cuda:0"))
x = torch.randn(128, 3, 32, 32, device="cuda")
y = torch.randint(0, 100, (128,), device="cuda")
for _ in range(10):
    eng.train_step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    eng.train_step(x, y)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"host enqueue {host / 30 * 1e3:.2f} ms/step, with GPU drain {total / 30 * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    eng.train_step(x, y)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
