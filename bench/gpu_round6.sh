#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 900 python -u -m pytest tests -m gpu -q -s --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest gpu rc=$rc"; grep -E "passed|failed|cosine|logits err|^FAILED|Error" gpurun_out/pytest_gpu.log | tail -20
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch_size 256 > gpurun_out/bench_native_b256.log 2>&1
rc=$?; echo "native bench b256 rc=$rc"; tail -1 gpurun_out/bench_native_b256.log | cut -c1-300
if [ $rc -ne 0 ]; then tail -30 gpurun_out/bench_native_b256.log; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b256 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --batch_size 256 > $GRAFT_REPO_ROOT/gpurun_out/prof_b256.log 2>&1
echo "rocprof rc=$?"
exit 0
