#!/bin/bash
# Second GPU pass: re-check fixed kernels, native vs stock-PyTorch bench, rocprofv3 kernel stats.
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 300 python -u bench/kernel_check.py --quick --out gpurun_out/kc_quick.json > gpurun_out/kc_quick.log 2>&1
rc=$?; echo "kernel_check quick rc=$rc"; grep -E "^FAIL|cases" gpurun_out/kc_quick.log | cut -c1-200
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --batch_size 128 > gpurun_out/bench_native_b128.log 2>&1
rc=$?; echo "native bench b128 rc=$rc"; tail -1 gpurun_out/bench_native_b128.log | cut -c1-400
if [ $rc -ne 0 ]; then tail -20 gpurun_out/bench_native_b128.log; exit $rc; fi
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --batch_size 256 > gpurun_out/bench_native_b256.log 2>&1
rc=$?; echo "native bench b256 rc=$rc"; tail -1 gpurun_out/bench_native_b256.log | cut -c1-400
if [ $rc -ne 0 ]; then exit $rc; fi
PDT_PLAIN_TORCH=1 PDT_FORCE_REFERENCE_OPS=1 timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --batch_size 256 > gpurun_out/bench_plain_torch_b256.log 2>&1
rc=$?; echo "plain torch b256 rc=$rc"; tail -1 gpurun_out/bench_plain_torch_b256.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_native -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --batch_size 128 > $GRAFT_REPO_ROOT/gpurun_out/prof_native.log 2>&1
rc=$?; echo "rocprof rc=$rc"; find $GRAFT_REPO_ROOT/gpurun_out/prof_native -name "*stats*" | head
exit 0
