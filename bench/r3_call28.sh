#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread -k "layernorm or vit" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python -u - <<'PY' || exit 1
import sys, torch
sys.path.insert(0, "bench")
import kernel_check as kc
from pytorch_distributed_training_amd.ops import native
for M, D in ((25216, 768), (100864, 768), (100864, 1024)):
    x = (torch.randn(M, D, device="cuda") * 2 + 0.3).to(torch.bfloat16)
    dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    add = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    gamma = torch.rand(D, device="cuda") + 0.5
    y, mean, rstd = native.layernorm_fwd(x, gamma, gamma, 1e-6)
    for a, nb in ((None, 6), (add, 8)):
        us = kc.timeit(lambda: native.layernorm_bwd(dy, x, mean, rstd, gamma, a))
        print(f"ln_bwd M={M} D={D} addend={a is not None}: {us:.1f} us  {nb * x.numel() / us / 1e6:.2f} TB/s", flush=True)
    dx, dg, db = native.layernorm_bwd(dy, x, mean, rstd, gamma, add)
    dx0, _, _ = native.layernorm_bwd(dy, x, mean, rstd, gamma)
    print("addend err", float((dx.float() - (dx0.float() + add.float())).abs().max()))
PY
echo "== vit B=512"
timeout -k 10 300 python -u bench.py --model vit_b_16 --optimizer adamw --batch_size 512 --steps 15 --warmup 4 2>/dev/null | cut -c1-180 || exit 1
