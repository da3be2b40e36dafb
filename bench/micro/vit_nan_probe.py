"""Where does the first non-finite value of a diverging ViT run come from?  Replays bench/convergence.py's ViT-B/16 run
(AdamW 3e-4, seed 0: goes to NaN between steps 250 and 300) and, at the first step whose loss or parameters are not
finite, reports per module the largest input / output magnitudes of the forward that produced it and the parameters /
gradients that were non-finite after the previous step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convergence import make_batches
from pytorch_distributed_training_amd import dist as pdist
from pytorch_distributed_training_amd.engine import Engine, EngineConfig

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
pdist.init_process_group(backend="rccl", local_rank=0)
eng = Engine(EngineConfig(model="vit_b_16", num_classes=10, dtype="bf16", optimizer="adamw", lr=3e-4, sync_bn=True, seed=0, image_size=224))
sample = make_batches(10, 224, seed=1)
gen = torch.Generator().manual_seed(2)
stats = {}
def hook(name):
    def f(m, inp, out):
        i = inp[0] if isinstance(inp, tuple) and len(inp) else inp
        o = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(i) and torch.is_tensor(o):
            stats[name] = (float(i.float().abs().max()), float(o.float().abs().max()), bool(torch.isfinite(o.float()).all()))
    return f
from pytorch_distributed_training_amd.ops import native, reference
from pytorch_distributed_training_amd import _C
_orig_bwd = native.attention_bwd
caught = {}
def checked_bwd(dout, qkv, out, lse, heads):
    d = _orig_bwd(dout, qkv, out, lse, heads)
    if not caught and not bool(torch.isfinite(d.float()).all()):
        caught["x"] = True
        fin = lambda t: bool(torch.isfinite(t.float()).all())
        B, S, D3 = qkv.shape
        D = D3 // 3
        q, k, v = (qkv[..., i * D:(i + 1) * D].float() for i in range(3))
        print(f"   [attention_bwd produced non-finite values] inputs finite: dout {fin(dout)} qkv {fin(qkv)} out {fin(out)} lse {fin(lse)}")
        print(f"   max|q| {float(q.abs().max()):.3e} max|k| {float(k.abs().max()):.3e} max|v| {float(v.abs().max()):.3e} max|dout| {float(dout.float().abs().max()):.3e} "
              f"max|out| {float(out.float().abs().max()):.3e} lse range [{float(lse.min()):.3e}, {float(lse.max()):.3e}]")
        bad = ~torch.isfinite(d.float())
        idx = bad.nonzero()
        print(f"   non-finite outputs: {int(bad.sum())} of {d.numel()}; images {sorted(set(idx[:, 0].tolist()))[:10]} rows {sorted(set(idx[:, 1].tolist()))[:12]} "
              f"column range [{int(idx[:, 2].min())}, {int(idx[:, 2].max())}] (dq < {D} <= dk < {2 * D} <= dv)")
        cols = idx[:, 2]
        print(f"   non-finite by part: dq {int((cols < D).sum())} dk {int(((cols >= D) & (cols < 2 * D)).sum())} dv {int((cols >= 2 * D).sum())}; heads {sorted(set(((cols % D) // 64).tolist()))}")
        _C.set_attention_pipelined(False)
        d0 = _orig_bwd(dout, qkv, out, lse, heads)
        _C.set_attention_pipelined(True)
        print(f"   one-head-per-workgroup kernels on the same inputs: finite {fin(d0)}")
        d1 = _orig_bwd(dout, qkv, out, lse, heads)
        print(f"   pipelined kernels again on the same inputs: finite {fin(d1)}")
        outr, lser = reference.attention_fwd(qkv.float(), heads)
        dr = reference.attention_bwd(dout.float(), qkv.float(), outr, lser, heads)
        print(f"   fp32 reference finite {fin(dr)} max|dqkv| {float(dr.abs().max()):.3e}; lse vs fp32 max diff {float((lse - lser).abs().max()):.3e}")
        if fin(d0):
            print(f"   one-head vs fp32: max err / max {float((d0.float() - dr).abs().max() / dr.abs().max()):.3e}")
    return d
native.attention_bwd = checked_bwd
model = eng.net.module if hasattr(eng.net, "module") else eng.net
names = {m: n for n, m in model.named_modules()}
prev_gnorm = None
for step in range(400):
    x, y = sample(64, gen, 0.1)
    hooks = []
    if step >= 200:
        stats.clear()
        hooks = [m.register_forward_hook(hook(n)) for m, n in names.items() if n and n.count(".") <= 3]
    loss = float(eng.train_step(x.to(eng.device), y.to(eng.device)))
    for h in hooks: h.remove()
    bad_p = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p.float()).all())]
    gmax = max((float(p.grad.float().abs().max()) for p in model.parameters() if p.grad is not None), default=0.0)
    if step % 25 == 0 or loss != loss or bad_p:
        print(f"step {step} loss {loss:.4f} max|grad| {gmax:.3e} non-finite params {len(bad_p)}", flush=True)
    if loss != loss or bad_p:
        big = sorted(stats.items(), key=lambda kv: -(kv[1][1] if kv[1][1] == kv[1][1] else 1e99))[:12]
        for n, (i, o, fin) in big:
            print(f"   {n:45s} max|in| {i:.3e} max|out| {o:.3e} finite {fin}")
        firstbad = [n for n, v in stats.items() if not v[2]]
        print("   first modules with non-finite output:", firstbad[:6])
        print("   non-finite params:", bad_p[:8])
        bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad.float()).all())]
        print("   non-finite grads:", bad_g[:8])
        break
