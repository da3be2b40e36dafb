#!/bin/bash
# persistent pipelined whole-head attention: tests, kernel timing old vs new, ViT step
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread -k "attention" > gpurun_out/r3/attn_tests.log 2>&1
rc=$?; tail -15 gpurun_out/r3/attn_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u - <<'PY' || exit 1
import sys, torch
sys.path.insert(0, "bench")
import kernel_check as kc
from pytorch_distributed_training_amd import _C
for on in (False, True):
    _C.set_attention_pipelined(on)
    for shape in ((128, 197, 12), (512, 197, 12)):
        kc.check_attention(*shape, True)
    print("pipelined", on)
    for r in kc.RESULTS[-4:]:
        print(r)
PY
echo "== vit B=128"
timeout -k 10 300 python -u bench.py --model vit_b_16 --optimizer adamw --batch_size 128 --steps 20 --warmup 5 2>/dev/null | cut -c1-180 || exit 1
echo "== vit B=512"
timeout -k 10 300 python -u bench.py --model vit_b_16 --optimizer adamw --batch_size 512 --steps 15 --warmup 4 2>/dev/null | cut -c1-180 || exit 1
