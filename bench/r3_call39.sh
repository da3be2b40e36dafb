#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R PDT_QUIET=1
run() { name=$1; shift; timeout -k 10 300 python -u bench.py "$@" 2>gpurun_out/r3/odd_$name.err | grep '^{' > gpurun_out/r3/odd_$name.json; rc=$?; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3/odd_$name.json"))
    print("$name", d["value"], "img/s", d["ms_per_step"], "ms  B", d["config"]["per_gpu_batch"], "fallbacks", d["config"]["library_fallbacks"], "loss", round(d["config"]["final_loss"], 3))
except Exception as e:
    print("$name FAILED rc=$rc", e); print(open("gpurun_out/r3/odd_$name.err").read()[-1500:])
PY
}
run r50_160 --image_size 160 --batch_size 250 --steps 6 --warmup 3
run r50_288 --image_size 288 --batch_size 200 --steps 6 --warmup 3
run r50_b37 --batch_size 37 --steps 6 --warmup 3
run r50_c1001 --num_classes 1001 --batch_size 64 --steps 6 --warmup 3
run vit_160 --model vit_b_16 --optimizer adamw --image_size 160 --batch_size 100 --steps 6 --warmup 3
run vit_256 --model vit_b_16 --optimizer adamw --image_size 256 --batch_size 60 --steps 6 --warmup 3
run vit_b37 --model vit_b_16 --optimizer adamw --batch_size 37 --steps 6 --warmup 3
run vit_ti_b300 --model vit_ti_16 --optimizer adamw --batch_size 300 --steps 6 --warmup 3
run r18_cifar_graph --model resnet18 --image_size 32 --num_classes 10 --batch_size 128 --optimizer adam --sync_bn --graph --steps 30 --warmup 10
