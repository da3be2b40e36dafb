#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread -k "attention" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python -u - <<'PY' || exit 1
import sys, torch
sys.path.insert(0, "bench")
import kernel_check as kc
for shape in ((32, 577, 12, 64), (64, 197, 6, 128), (64, 197, 24, 32), (16, 1024, 12, 64)):
    kc.check_attention(shape[0], shape[1], shape[2], True, hd=shape[3])
for r in kc.RESULTS[-8:]:
    print(r["case"], r["ok"], r.get("us"))
PY
