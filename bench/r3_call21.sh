#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "layernorm or vit" 2>&1 | tail -4
timeout -k 10 300 python -u - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench.kernel_check as kc
for shape in ((25216, 768), (25216, 192), (25216, 1024), (25216, 1280), (100864, 768)):
    kc.check_layernorm(*shape, True)
for r in kc.RESULTS[-10:]:
    print(r)
PY
timeout -k 10 300 python -u bench.py --model vit_b_16 --optimizer adamw --batch_size 128 --steps 20 --warmup 5 2>/dev/null | cut -c1-200
