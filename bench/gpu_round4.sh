#!/bin/bash
# pytest -m gpu, then an igemm config sweep (PDT_IGEMM_CFG) over the ResNet-50 layer shapes.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest gpu rc=$rc"; tail -15 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
for cfg in 0 1 2 3 4; do
  PDT_IGEMM_CFG=$cfg timeout -k 10 200 python -u bench/kernel_check.py --time --only conv --out gpurun_out/sweep_cfg$cfg.json > gpurun_out/sweep_cfg$cfg.log 2>&1
  rc=$?; echo "sweep cfg $cfg rc=$rc"; grep -E "cases" gpurun_out/sweep_cfg$cfg.log
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
done
exit 0
