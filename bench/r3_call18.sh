#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "both_norms or resnet or two_norms or oracle" > gpurun_out/r3/pytest_dual.log 2>&1
rc=$?; tail -25 gpurun_out/r3/pytest_dual.log | cut -c1-400; [ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0; do
  PDT_FUSE_DUAL_BN_SUMS=$v timeout -k 10 300 python -u bench.py --steps 12 --warmup 4 --batch_size 1280 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dual_sums=$v', d['value'], d['ms_per_step'])" || exit 1
done
