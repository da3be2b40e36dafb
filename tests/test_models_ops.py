"""T0/T1 on CPU: model parity (state_dict keys, parameter counts), fused autograd nodes vs stock
PyTorch modules, optimizers vs torch.optim, checkpoint format, profiler schedule, log format."""
import math
import re

import pytest
import torch
import torch.nn as tnn

import pytorch_distributed_training_amd as pdt
from pytorch_distributed_training_amd import nn as pnn
from pytorch_distributed_training_amd.models import build_model, resnet18, resnet50
from pytorch_distributed_training_amd.nn import functional as F
from pytorch_distributed_training_amd.optim import SGD, Adam, AdamW
from pytorch_distributed_training_amd.utils import metrics, profiling


# ---------------------------------------------------------------------------- models
def test_resnet50_matches_torchvision_contract():
    """25 557 032 parameters in 161 tensors, 159 buffers, torchvision key names."""
    m = resnet50()
    params = dict(m.named_parameters())
    assert sum(p.numel() for p in params.values()) == 25_557_032
    assert len(params) == 161
    assert len(list(m.buffers())) == 159
    keys = set(m.state_dict().keys())
    for k in ("conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var",
              "bn1.num_batches_tracked", "layer1.0.conv1.weight", "layer1.0.bn3.weight",
              "layer1.0.downsample.0.weight", "layer1.0.downsample.1.running_var", "layer4.2.conv3.weight",
              "fc.weight", "fc.bias"):
        assert k in keys, k
    pat = re.compile(r"^(conv1|bn1|fc|layer[1-4]\.\d+\.(conv[1-3]|bn[1-3]|downsample\.[01]))\.")
    assert all(pat.match(k) for k in keys)
    assert params["conv1.weight"].shape == (64, 3, 7, 7)
    assert params["layer2.0.conv2.weight"].shape == (128, 128, 3, 3) and m.layer2[0].conv2.stride == (2, 2)  # v1.5
    assert params["layer2.0.conv1.weight"].shape == (128, 256, 1, 1) and m.layer2[0].conv1.stride == (1, 1)
    assert params["fc.weight"].shape == (1000, 2048)


@pytest.mark.parametrize("name,count", [("resnet18", 11_689_512), ("resnet34", 21_797_672), ("resnet101", 44_549_160),
                                        ("resnet152", 60_192_808), ("vit_b_16", 86_567_656), ("vit_ti_16", 5_717_416),
                                        ("vit_s_16", 22_050_664), ("vit_l_16", 304_326_632)])
def test_parameter_counts(name, count):
    with torch.device("meta"):  # (shapes only: ViT-L/16 is 1.2 GB of fp32)
        model = build_model(name)
    assert sum(p.numel() for p in model.parameters()) == count


def test_vit_state_dict_keys():
    keys = set(build_model("vit_b_16").state_dict().keys())
    for k in ("class_token", "conv_proj.weight", "conv_proj.bias", "encoder.pos_embedding",
              "encoder.layers.encoder_layer_0.ln_1.weight", "encoder.layers.encoder_layer_0.self_attention.in_proj_weight",
              "encoder.layers.encoder_layer_0.self_attention.in_proj_bias",
              "encoder.layers.encoder_layer_0.self_attention.out_proj.weight",
              "encoder.layers.encoder_layer_11.mlp.0.weight", "encoder.layers.encoder_layer_11.mlp.3.bias",
              "encoder.ln.weight", "heads.head.weight", "heads.head.bias"):
        assert k in keys, k
    assert len(keys) == 152


def _stock_resnet18_like(ours):
    """A plain torch.nn module tree with identical keys, for a forward/backward oracle."""
    class Block(tnn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.conv1 = tnn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = tnn.BatchNorm2d(cout)
            self.conv2 = tnn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = tnn.BatchNorm2d(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = tnn.Sequential(tnn.Conv2d(cin, cout, 1, stride, bias=False), tnn.BatchNorm2d(cout))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = torch.relu(self.bn1(self.conv1(x)))
            return torch.relu(self.bn2(self.conv2(out)) + idt)

    class Net(tnn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = tnn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = tnn.BatchNorm2d(64)
            self.layer1 = tnn.Sequential(Block(64, 64, 1), Block(64, 64, 1))
            self.layer2 = tnn.Sequential(Block(64, 128, 2), Block(128, 128, 1))
            self.layer3 = tnn.Sequential(Block(128, 256, 2), Block(256, 256, 1))
            self.layer4 = tnn.Sequential(Block(256, 512, 2), Block(512, 512, 1))
            self.fc = tnn.Linear(512, 10)

        def forward(self, x):
            x = tnn.functional.max_pool2d(torch.relu(self.bn1(self.conv1(x))), 3, 2, 1)
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
            return self.fc(x.mean((2, 3)))

    net = Net()
    net.load_state_dict(ours.state_dict(), strict=True)  # key-for-key interchange, strict
    return net


def test_resnet18_forward_backward_matches_stock_modules():
    torch.manual_seed(0)
    ours = resnet18(num_classes=10)
    ref = _stock_resnet18_like(ours)
    x = torch.randn(4, 3, 32, 32)
    y = torch.randint(0, 10, (4,))
    lo = pnn.CrossEntropyLoss()(ours(x), y)
    lr = tnn.functional.cross_entropy(ref(x), y)
    assert torch.allclose(lo, lr, atol=1e-5)
    lo.backward()
    lr.backward()
    for (n, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        scale = float(q.grad.abs().max()) + 1e-6  # BN over 4 samples at 1x1 spatial is ill-conditioned
        assert float((p.grad - q.grad).abs().max()) / scale < 2e-3, n
    for (n, b), (_, c) in zip(ours.named_buffers(), ref.named_buffers()):
        assert torch.allclose(b.float(), c.float(), atol=1e-5), n
    # eval mode uses the running statistics
    ours.eval(); ref.eval()
    assert torch.allclose(ours(x), ref(x), atol=1e-4)


def test_vit_tiny_trains():
    torch.manual_seed(0)
    m = build_model("vit_tiny_test", num_classes=10)
    opt = AdamW(m.parameters(), lr=1e-2)
    x, y = torch.randn(8, 3, 32, 32), torch.randint(0, 10, (8,))
    losses = []
    for _ in range(12):
        loss = pnn.CrossEntropyLoss()(m(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] * 0.7


# ------------------------------------------------------------------- fused nodes (CPU)
@pytest.mark.parametrize("relu,use_res", [(True, True), (True, False), (False, True), (False, False)])
def test_bn_act_node_matches_stock_ops(relu, use_res):
    torch.manual_seed(1)
    bn = pnn.BatchNorm2d(6)
    ref = tnn.BatchNorm2d(6)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
    ref.load_state_dict(bn.state_dict())
    x = torch.randn(3, 6, 5, 4, requires_grad=True)
    r = torch.randn(3, 6, 5, 4, requires_grad=True)
    x2, r2 = x.detach().clone().requires_grad_(), r.detach().clone().requires_grad_()
    out = bn(x, residual=r if use_res else None, relu=relu)
    o2 = ref(x2) + (r2 if use_res else 0)
    o2 = torch.relu(o2) if relu else o2
    assert torch.allclose(out, o2, atol=1e-5)
    g = torch.randn_like(out)
    out.backward(g); o2.backward(g)
    assert torch.allclose(x.grad, x2.grad, atol=1e-5)
    if use_res:
        assert torch.allclose(r.grad, r2.grad, atol=1e-6)
    assert torch.allclose(bn.weight.grad, ref.weight.grad, atol=1e-4)
    assert torch.allclose(bn.bias.grad, ref.bias.grad, atol=1e-4)
    assert torch.allclose(bn.running_var, ref.running_var, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1


def test_softmax_ce_node():
    torch.manual_seed(2)
    logits = torch.randn(7, 13, requires_grad=True)
    l2 = logits.detach().clone().requires_grad_()
    y = torch.randint(0, 13, (7,))
    a = F.softmax_cross_entropy(logits, y)
    b = tnn.functional.cross_entropy(l2, y)
    (a * 3).backward(); (b * 3).backward()
    assert torch.allclose(a, b, atol=1e-6) and torch.allclose(logits.grad, l2.grad, atol=1e-6)


def test_convert_sync_batchnorm_shares_parameters():
    """The reference builds Adam BEFORE converting (main.py:80,82): Parameter identity must survive."""
    net = resnet18(num_classes=10)
    before = {n: p for n, p in net.named_parameters()}
    bufs = {n: b for n, b in net.named_buffers()}
    conv = pnn.convert_sync_batchnorm(net)
    after = dict(conv.named_parameters())
    assert set(before) == set(after) and all(before[n] is after[n] for n in before)
    assert all(bufs[n] is b for n, b in conv.named_buffers())
    assert all(isinstance(m, pnn.SyncBatchNorm) for m in conv.modules() if isinstance(m, tnn.modules.batchnorm._BatchNorm))
    assert set(conv.state_dict()) == set(resnet18(num_classes=10).state_dict())
    # single process: identical to plain BN
    x = torch.randn(2, 3, 32, 32)
    assert conv(x).shape == (2, 10)


# ----------------------------------------------------------------------- optimizers
def _pair(shapes, seed=0):
    g = torch.Generator().manual_seed(seed)
    a = [torch.randn(s, generator=g).requires_grad_() for s in shapes]
    b = [t.detach().clone().requires_grad_() for t in a]
    return a, b


@pytest.mark.parametrize("kind", ["adam", "adamw", "sgd", "sgd_nesterov", "sgd_plain"])
def test_optimizers_match_torch(kind):
    shapes = [(5, 3), (7,), (2, 3, 3, 3)]
    a, b = _pair(shapes)
    mk = {
        "adam": (lambda p: Adam(p, lr=1e-2, weight_decay=0.1), lambda p: torch.optim.Adam(p, lr=1e-2, weight_decay=0.1)),
        "adamw": (lambda p: AdamW(p, lr=1e-2, weight_decay=0.1), lambda p: torch.optim.AdamW(p, lr=1e-2, weight_decay=0.1)),
        "sgd": (lambda p: SGD(p, lr=0.1, momentum=0.9, weight_decay=1e-2), lambda p: torch.optim.SGD(p, lr=0.1, momentum=0.9, weight_decay=1e-2)),
        "sgd_nesterov": (lambda p: SGD(p, lr=0.1, momentum=0.9, nesterov=True), lambda p: torch.optim.SGD(p, lr=0.1, momentum=0.9, nesterov=True)),
        "sgd_plain": (lambda p: SGD(p, lr=0.1), lambda p: torch.optim.SGD(p, lr=0.1)),
    }[kind]
    oa, ob = mk[0](a), mk[1](b)
    g = torch.Generator().manual_seed(5)
    for _ in range(4):
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g)
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, atol=1e-6, rtol=1e-5)


def test_optimizer_state_dict_interchanges_with_torch():
    a, b = _pair([(4, 4), (3,)])
    oa, ob = Adam(a, lr=1e-2), torch.optim.Adam(b, lr=1e-2)
    for p, q in zip(a, b):
        p.grad = torch.ones_like(p); q.grad = torch.ones_like(q)
    oa.step(); ob.step()
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    import copy

    ob.load_state_dict(copy.deepcopy(sd))               # ours -> torch
    oa.load_state_dict(copy.deepcopy(ob.state_dict()))  # torch -> ours (deepcopy: no shared `step` tensor)
    for p, q in zip(a, b):
        p.grad = torch.full_like(p, 0.5); q.grad = torch.full_like(q, 0.5)
    oa.step(); ob.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, atol=1e-6)


def test_optimizer_built_before_flattening_keeps_working():
    """main.py:79-83 order: optimizer first, then device move / DDP flattening."""
    from pytorch_distributed_training_amd.parallel import ParamArena

    net = resnet18(num_classes=10)
    opt = SGD(net.parameters(), lr=0.1, momentum=0.9)
    arena = ParamArena(net)
    flat = arena.by_dtype[torch.float32]
    before = flat.clone()
    loss = net(torch.randn(2, 3, 32, 32)).sum()
    loss.backward()
    opt.step()
    assert not torch.equal(flat, before)  # the step wrote into the arena through the same Parameters
    assert all(p.data_ptr() >= flat.data_ptr() and p.data_ptr() < flat.data_ptr() + flat.numel() * 4
               for p in net.parameters())


# ----------------------------------------------------------------- metrics / profiler
def test_log_format_matches_reference():
    assert metrics.HEADER == "datetime\tg_step\tg_img\tloss_value\texamples_per_sec\n"
    assert metrics.log_file_name("Job0", 128, 3) == "Job0_128_3.log"
    row = metrics.format_row("2026-01-01 00:00:00.000001", 5, 8, 128, 6.9, 1234.5)
    assert row == "2026-01-01 00:00:00.000001\t40\t5120\t6.9\t1234.5\n"  # g_step = step*W, g_img = step*W*B


def test_throughput_log_cadence(tmp_path):
    log = metrics.ThroughputLog("J", 16, 0, 2, str(tmp_path))
    assert [s for s in range(1, 21) if log.should_log(s)] == [5, 10, 15, 20]
    assert not metrics.ThroughputLog("J", 16, 1, 2, str(tmp_path)).should_log(5)  # only rank 0 writes rows
    log.row(5, 1.5, 0.25)
    log.footer(); log.close()
    lines = open(tmp_path / "J_16_0.log").read().splitlines()
    assert lines[0] == metrics.HEADER.strip("\n")
    f = lines[1].split("\t")
    assert f[1:4] == ["10", "160", "1.5"] and float(f[4]) == 64.0
    assert re.match(r"^TrainTime\t\d+\.\d{6}$", lines[2])


def test_profiler_schedule_matches_torch():
    import torch.profiler as tp

    sched = tp.schedule(**profiling.SCHEDULE)
    for step in range(20):
        assert sched(step).name == profiling.schedule_action(step), step
    assert [profiling.schedule_action(s) for s in (0, 1, 2, 3, 4, 8, 9, 10)] == [
        "NONE", "NONE", "WARMUP", "WARMUP", "RECORD", "RECORD", "RECORD_AND_SAVE", "NONE"]


def test_profiler_modes(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    assert isinstance(profiling.make_profiler("none", "J"), profiling.NullProfiler)
    with profiling.make_profiler("torch", "J", with_stack=False) as p:
        for _ in range(11):
            torch.ones(4).sum()
            p.step()
    traces = list((tmp_path / "log_J").glob("*.pt.trace.json"))
    assert len(traces) == 1
    with pytest.raises(ValueError):
        profiling.make_profiler("bogus", "J")
    with profiling.range_("noop"):
        pass


def test_native_extension_builds_and_imports():
    """The in-tree extension must import even without a GPU (hipcc cross-compiles gfx950)."""
    if not pdt.native_available():
        pytest.skip("extension not built in this checkout (run python __graft_entry__.py build)")
    ext = pdt.require_native()
    assert ext.arch == "gfx950"
    for name in ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad", "bn_finalize", "softmax_ce", "multi_adam", "Reducer",
                 "RcclComm"):
        assert hasattr(ext, name)
    plan = ext.plan_buckets([10, 300, 5], [2, 1, 0], 4, 1 << 20, 1 << 10)
    assert [p[0] for p in plan] == [[2], [1, 0]] or [p[0] for p in plan] == [[2, 1], [0]] or len(plan) >= 1


def test_layer_norm_function_matches_torch_autograd():
    from pytorch_distributed_training_amd.nn.functional import layer_norm

    torch.manual_seed(0)
    x = torch.randn(3, 7, 32, requires_grad=True)
    w = torch.rand(32, requires_grad=True)
    b = torch.randn(32, requires_grad=True)
    dy = torch.randn(3, 7, 32)
    y = layer_norm(x, w, b, 1e-6)
    y.backward(dy)
    got = [y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    for t in (x, w, b):
        t.grad = None
    yr = torch.nn.functional.layer_norm(x, (32,), w, b, 1e-6)
    yr.backward(dy)
    for a, r in zip(got, [yr.detach(), x.grad, w.grad, b.grad]):
        assert torch.allclose(a, r, rtol=1e-4, atol=1e-5)


def test_attention_function_matches_torch_autograd():
    """The oracle's hand-written attention backward (what the HIP kernel is checked against) vs autograd."""
    from pytorch_distributed_training_amd.nn.functional import attention

    torch.manual_seed(0)
    b, s, h, hd = 2, 5, 3, 8
    qkv = torch.randn(b, s, 3 * h * hd, requires_grad=True)
    dy = torch.randn(b, s, h * hd)
    out = attention(qkv, h)
    out.backward(dy)
    g = qkv.grad.clone()
    qkv.grad = None
    q, k, v = qkv.view(b, s, 3, h, hd).permute(2, 0, 3, 1, 4)
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, s, h * hd)
    ref.backward(dy)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
    assert torch.allclose(g, qkv.grad, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ lazily gated residual gradient
@pytest.mark.parametrize("name", ["resnet18", "resnet50"])
def test_lazy_residual_grad_matches_materialised(name, monkeypatch):
    """The block-output BN hands the residual's producer the UNGATED gradient + its ReLU mask
    (functional._tag_gate); gradients must equal the path that writes the gated tensor
    (reference semantics: /root/reference/main.py:100-103, autograd of relu(bn(y) + identity))."""
    import pytorch_distributed_training_amd.models.resnet as R
    import pytorch_distributed_training_amd.nn.functional as F

    def grads(lazy):
        monkeypatch.setattr(R, "_LAZY_RES", lazy)
        tags = []
        orig = F._tag_gate
        monkeypatch.setattr(F, "_tag_gate", lambda dz, m: (tags.append(1), orig(dz, m))[1])
        torch.manual_seed(0)
        from pytorch_distributed_training_amd.models import build_model
        m = build_model(name, 10)
        x = torch.randn(2, 3, 32, 32, requires_grad=True)
        m(x).square().sum().backward()
        return len(tags), x.grad.clone(), [p.grad.clone() for p in m.parameters()]

    n1, gx1, gp1 = grads(True)
    n0, gx0, gp0 = grads(False)
    assert n1 == {"resnet18": 8, "resnet50": 16}[name] and n0 == 0
    assert torch.equal(gx1, gx0)
    for a, b in zip(gp1, gp0):
        assert torch.equal(a, b)


def test_lazy_residual_grad_refuses_second_consumer():
    """A residual with two consumers would receive an ungated gradient through autograd's sum: the producer was
    told to expect the gate and must raise instead of training on a wrong gradient."""
    import pytorch_distributed_training_amd.nn as pnn
    from pytorch_distributed_training_amd.nn.functional import conv_bn_act

    torch.manual_seed(0)
    c1, b1 = pnn.Conv2d(8, 8, 1, bias=False), pnn.BatchNorm2d(8)
    c2, b2 = pnn.Conv2d(8, 8, 3, padding=1, bias=False), pnn.BatchNorm2d(8)
    x = torch.randn(2, 8, 6, 6, requires_grad=True)
    out, skip = conv_bn_act(x, c1, b1, relu=True, return_skip=True)
    z = conv_bn_act(out, c2, b2, relu=True, residual=skip, lazy_residual_grad=True)
    with pytest.raises(RuntimeError, match="lazy residual gradient"):
        (z.sum() + skip.sum()).backward()


# ------------------------------------------------------------------ deferred downsample BN apply
@pytest.mark.parametrize("train", [True, False])
def test_deferred_downsample_bn_matches_materialised(train, monkeypatch):
    """The downsample BN returns its input un-normalised and the block-output node applies its affine while reading
    the residual (functional._PendingAffine): outputs, gradients and running statistics must not change."""
    import pytorch_distributed_training_amd.models.resnet as R
    import pytorch_distributed_training_amd.nn.functional as F
    from pytorch_distributed_training_amd.models import build_model

    def run(defer):
        monkeypatch.setattr(R, "_DEFER_DS", defer)
        made = []
        orig = F._PendingAffine.__init__
        monkeypatch.setattr(F._PendingAffine, "__init__", lambda self: (made.append(1), orig(self))[1])
        torch.manual_seed(0)
        m = build_model("resnet50", 10)
        m.train(train)
        x = torch.randn(2, 3, 32, 32, requires_grad=True)
        out = m(x)
        out.square().sum().backward()
        return len(made), out.detach(), x.grad.clone(), [p.grad.clone() for p in m.parameters()], \
            [b.clone() for b in m.buffers()]

    n1, o1, gx1, gp1, b1 = run(True)
    n0, o0, gx0, gp0, b0 = run(False)
    assert n1 == 4 and n0 == 0  # one downsample branch per stage
    assert torch.allclose(o1, o0, rtol=1e-5, atol=1e-6) and torch.allclose(gx1, gx0, rtol=1e-4, atol=1e-6)
    for a, b in zip(gp1, gp0):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    for a, b in zip(b1, b0):
        assert torch.allclose(a.float(), b.float(), rtol=1e-6, atol=1e-7)


def test_downsample_bn_backward_shares_the_pass_of_the_block_output_bn(monkeypatch):
    """relu(bn3(x) + bn_ds(x_ds)): both BNs receive the same gated gradient, so the block-output node computes both
    input gradients at once (bn_bwd_elemt2) and the downsample BN's backward only hands its share on."""
    import pytorch_distributed_training_amd.nn.functional as F
    import pytorch_distributed_training_amd.ops.reference as ref
    from pytorch_distributed_training_amd.models import build_model

    def run(dual):
        monkeypatch.setattr(F, "DUAL_BN_BACKWARD", dual)
        calls = []
        orig = ref.bn_bwd_elemt2
        monkeypatch.setattr(ref, "bn_bwd_elemt2", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        torch.manual_seed(0)
        m = build_model("resnet50", 10)
        x = torch.randn(2, 3, 32, 32, requires_grad=True)
        m(x).square().sum().backward()
        monkeypatch.setattr(ref, "bn_bwd_elemt2", orig)
        return len(calls), x.grad.clone(), [p.grad.clone() for p in m.parameters()]

    n1, gx1, gp1 = run(True)
    n0, gx0, gp0 = run(False)
    assert n1 == 4 and n0 == 0  # one downsample branch per stage
    assert torch.allclose(gx1, gx0, rtol=1e-5, atol=1e-7)
    for a, b in zip(gp1, gp0):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_deferred_bn_refuses_unaware_consumer():
    import pytorch_distributed_training_amd.nn as pnn
    from pytorch_distributed_training_amd.nn.functional import conv_bn_act

    c, b = pnn.Conv2d(8, 8, 1, bias=False), pnn.BatchNorm2d(8)
    x = torch.randn(2, 8, 4, 4, requires_grad=True)
    y = conv_bn_act(x, c, b, relu=False, defer_apply=True)
    with pytest.raises(RuntimeError, match="deferred BatchNorm apply"):
        y.sum().backward()


def test_compact_downsample_grad_matches_dense(monkeypatch):
    """The 1x1 stride-2 downsample conv hands its data gradient to the block's first conv compact
    ([B, C, H/2, W/2], side channel on the producer's ctx) instead of zero-filled at full size."""
    import pytorch_distributed_training_amd.models.resnet as R
    import pytorch_distributed_training_amd.ops.reference as ref
    from pytorch_distributed_training_amd.models import build_model

    def run(flag):
        monkeypatch.setattr(R, "_COMPACT_DS", flag)
        used = []
        orig = ref.conv2d_dgrad

        def spy(*a, **k):
            if (len(a) > 8 and a[8] == 2) or k.get("addend_stride", 1) == 2:
                used.append(1)
            return orig(*a, **k)
        monkeypatch.setattr(ref, "conv2d_dgrad", spy)
        torch.manual_seed(0)
        m = build_model("resnet50", 10)
        x = torch.randn(2, 3, 32, 32, requires_grad=True)
        m(x).square().sum().backward()
        return len(used), x.grad.clone(), [p.grad.clone() for p in m.parameters()]

    n1, gx1, gp1 = run(True)
    n0, gx0, gp0 = run(False)
    assert n1 == 3 and n0 == 0  # layer2/3/4 open with a stride-2 downsample

    def close(a, b):  # same sums in a different order: compare against the tensor's scale
        return float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12

    assert close(gx1, gx0)
    for a, b in zip(gp1, gp0):
        assert close(a, b)
