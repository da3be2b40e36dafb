#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest gpu rc=$rc"; grep -E "passed|failed|^FAILED|Error" gpurun_out/pytest_gpu.log | tail -10
if [ $rc -ne 0 ]; then tail -30 gpurun_out/pytest_gpu.log; exit $rc; fi
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?: $(tail -1 gpurun_out/smoke.log)"
timeout -k 10 400 python -u bench.py --steps 8 --warmup 3 --batch_size 0 > gpurun_out/bench_auto.log 2>&1
echo "auto rc=$?: $(tail -1 gpurun_out/bench_auto.log | cut -c1-400)"
timeout -k 10 300 python -u bench.py --steps 10 --warmup 4 > gpurun_out/bench_b1024.log 2>&1
echo "b1024 rc=$?: $(tail -1 gpurun_out/bench_b1024.log | cut -c1-200)"
exit 0
