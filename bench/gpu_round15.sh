#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 400 python -u bench/kernel_check.py --time --only conv > gpurun_out/kc_conv.log 2>&1
rc=$?; echo "conv check rc=$rc"; grep -E "FAIL|bn_reduce|cases," gpurun_out/kc_conv.log | cut -c1-200
if [ $rc -gt 1 ]; then tail -20 gpurun_out/kc_conv.log; exit $rc; fi
if [ $rc -eq 1 ]; then exit 1; fi
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest gpu rc=$rc"; grep -E "passed|failed|^FAILED|Error" gpurun_out/pytest_gpu.log | tail -10
if [ $rc -ne 0 ]; then tail -30 gpurun_out/pytest_gpu.log; exit $rc; fi
for fuse in 0 1; do
PDT_FUSE_BN_REDUCE=$fuse timeout -k 10 300 python -u bench.py --steps 10 --warmup 4 > gpurun_out/bench_b1024_f$fuse.log 2>&1
echo "b1024 fuse=$fuse rc=$?: $(tail -1 gpurun_out/bench_b1024_f$fuse.log | cut -c1-200)"
PDT_FUSE_BN_REDUCE=$fuse timeout -k 10 300 python -u bench.py --steps 15 --warmup 5 --batch_size 256 > gpurun_out/bench_b256_f$fuse.log 2>&1
echo "b256 fuse=$fuse rc=$?: $(tail -1 gpurun_out/bench_b256_f$fuse.log | cut -c1-200)"
done
exit 0
