"""Which data-flow switch changes the native result, and by how much?  (diagnostic)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytorch_distributed_training_amd.models.resnet as R
import pytorch_distributed_training_amd.nn.functional as F
from pytorch_distributed_training_amd import nn as pnn
from pytorch_distributed_training_amd.models import build_model
from pytorch_distributed_training_amd.parallel import ParamArena

torch.manual_seed(0)
net = (R.ResNet(R.Bottleneck, [1, 1, 1, 1], num_classes=10) if os.environ.get("SMALL", "0") == "1"
       else build_model("resnet50", num_classes=10)).cuda()
arena = ParamArena(net); arena.attach_lowp(torch.bfloat16)
B, S = int(os.environ.get("B", 16)), int(os.environ.get("S", 64))
x = torch.randn(B, 3, S, S, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 10, (B,), device="cuda")
crit = pnn.CrossEntropyLoss()
KNOBS = [(R, "_LAZY_RES"), (R, "_DEFER_DS"), (R, "_COMPACT_DS"), (F, "_FUSE_STEM_POOL"), (F, "_FUSE_BN_REDUCE")]

def run(off=()):
    for mod, n in KNOBS:
        setattr(mod, n, n not in off)
    for p in net.parameters(): p.grad = None
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.reset_running_stats()
    out = net(x); crit(out, y).backward()
    return out.detach().float(), {n: p.grad.detach().float().clone() for n, p in net.named_parameters()}

def cmp(a, b):
    (o1, g1), (o0, g0) = a, b
    err = float((o1 - o0).abs().max() / o0.abs().max())
    cs = {n: float(torch.dot(g1[n].flatten(), g0[n].flatten()) / (g1[n].norm() * g0[n].norm() + 1e-12)) for n in g0}
    w = min(cs, key=cs.get)
    return err, cs[w], w

if os.environ.get("REF", "0") == "1":
    from pytorch_distributed_training_amd import ops
    ops.set_force_reference(True)
base = run()
print("repeat (determinism):", cmp(run(), base))
for _, n in KNOBS:
    print(f"{n} off:", cmp(run((n,)), base))
print("all off:", cmp(run(tuple(n for _, n in KNOBS)), base))
