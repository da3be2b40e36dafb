#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 500 python -u bench/resnet50_roofline.py --batch 1280 --out gpurun_out/r3/roofline_b1280_final.json > gpurun_out/r3/roofline_b1280_final.md 2>gpurun_out/r3/roofline.err || { tail -5 gpurun_out/r3/roofline.err; exit 1; }
tail -8 gpurun_out/r3/roofline_b1280_final.md
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/vit_prof_final -- python3 $R/bench.py --model vit_b_16 --optimizer adamw --batch_size 512 --steps 6 --warmup 2 > $R/gpurun_out/r3/vit_prof_final.log 2>&1 || exit 1
rm -f $R/gpurun_out/r3/vit_prof_final/*/*kernel_trace.csv
grep '^{' $R/gpurun_out/r3/vit_prof_final.log | cut -c1-170
