#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "captured_step_with_peer" > gpurun_out/r3/pytest_w2graph.log 2>&1
rc=$?; tail -30 gpurun_out/r3/pytest_w2graph.log | cut -c1-700; exit $rc
