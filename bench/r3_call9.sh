#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "two_shot or two_engine_ranks" > gpurun_out/r3/pytest_ipc2.log 2>&1
rc=$?; tail -40 gpurun_out/r3/pytest_ipc2.log | cut -c1-600; exit $rc
