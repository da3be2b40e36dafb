#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest gpu rc=$rc"; grep -E "passed|failed|^FAILED|Error" gpurun_out/pytest_gpu.log | tail -10
if [ $rc -ne 0 ]; then tail -30 gpurun_out/pytest_gpu.log; exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 10 --warmup 4 > gpurun_out/bench_b1024.log 2>&1
rc=$?; echo "b1024 rc=$rc: $(tail -1 gpurun_out/bench_b1024.log | cut -c1-200)"
[ $rc -eq 0 ] || { tail -20 gpurun_out/bench_b1024.log; exit $rc; }
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch_size 256 > gpurun_out/bench_b256.log 2>&1
rc=$?; echo "b256 rc=$rc: $(tail -1 gpurun_out/bench_b256.log | cut -c1-200)"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b1024 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_b1024.log 2>&1
echo "rocprof rc=$?"
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_b1024/*/*kernel_trace.csv
exit 0
