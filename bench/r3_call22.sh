#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 300 python -u - <<'PY'
import sys, torch
sys.path.insert(0, "bench")
import kernel_check as kc
for shape in ((25216, 768), (25216, 192), (25216, 1024), (25216, 1280), (100864, 768)):
    kc.check_layernorm(*shape, True)
for r in kc.RESULTS[-10:]:
    print(r)
PY
