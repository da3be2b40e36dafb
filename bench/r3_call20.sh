#!/bin/bash
# round 3: kernel-stats profiles of the tree (ResNet-50 default batch, B=1280) + per-layer roofline
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b2560 -- python3 $R/bench.py --steps 4 --warmup 2 > $R/gpurun_out/r3/prof_b2560.log 2>&1 || { tail -5 $R/gpurun_out/r3/prof_b2560.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1280 -- python3 $R/bench.py --steps 6 --warmup 3 --batch_size 1280 > $R/gpurun_out/r3/prof_b1280.log 2>&1 || { tail -5 $R/gpurun_out/r3/prof_b1280.log; exit 1; }
rm -f $R/gpurun_out/prof_b*/*/*kernel_trace.csv
cd $R
timeout -k 10 500 python -u bench/resnet50_roofline.py --batch 1280 > gpurun_out/r3/roofline_b1280.md 2>gpurun_out/r3/roofline_b1280.err || { tail -5 gpurun_out/r3/roofline_b1280.err; exit 1; }
tail -8 gpurun_out/r3/roofline_b1280.md
