#!/bin/bash
# the other BASELINE.json configurations and model families on the final tree
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R PDT_QUIET=1
run() { name=$1; shift; timeout -k 10 400 python -u bench.py "$@" 2>gpurun_out/r3/cfg_$name.err | grep '^{' > gpurun_out/r3/cfg_$name.json; rc=$?; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3/cfg_$name.json"))
    print("$name", d["value"], "img/s", d["ms_per_step"], "ms  B", d["config"]["per_gpu_batch"], "fallbacks", d["config"]["library_fallbacks"], "loss", round(d["config"]["final_loss"], 3))
except Exception as e:
    print("$name FAILED rc=$rc", e)
PY
}
run syncbn_b2560 --steps 8 --warmup 3 --sync_bn
run resnet152 --steps 6 --warmup 3 --model resnet152
run resnet152_b256 --steps 8 --warmup 3 --model resnet152 --batch_size 256
run resnet101_b512 --steps 8 --warmup 3 --model resnet101 --batch_size 512
run resnet34_b512 --steps 10 --warmup 4 --model resnet34 --batch_size 512
run resnet18_b512 --steps 10 --warmup 4 --model resnet18 --batch_size 512
run bf16_buckets --steps 8 --warmup 3 --grad_dtype bf16
run vit_l_16_b128 --steps 6 --warmup 3 --model vit_l_16 --optimizer adamw --batch_size 128
run vit_s_16_b512 --steps 8 --warmup 3 --model vit_s_16 --optimizer adamw --batch_size 512
run vit_b_16_384 --steps 6 --warmup 3 --model vit_b_16 --optimizer adamw --batch_size 64 --image_size 384
