#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 400 python -u bench/kernel_check.py --time --only conv > gpurun_out/kc_conv.log 2>&1
rc=$?; echo "conv check rc=$rc"; grep -E "FAIL|bn_reduce B128|cases," gpurun_out/kc_conv.log | cut -c1-200
if [ $rc -ne 0 ]; then tail -20 gpurun_out/kc_conv.log; exit $rc; fi
for fuse in 0 1; do
PDT_FUSE_BN_REDUCE=$fuse timeout -k 10 300 python -u bench.py --steps 10 --warmup 4 > gpurun_out/bench_b1024_f$fuse.log 2>&1
echo "b1024 fuse=$fuse rc=$?: $(tail -1 gpurun_out/bench_b1024_f$fuse.log | cut -c1-200)"
done
exit 0
