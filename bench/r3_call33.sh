#!/bin/bash
# convergence of the final kernels: ViT-B/16 (pipelined attention, LayerNorm, GEMM core) and ResNet-50 at 224 against the fp32 oracle
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r3 && export PYTHONPATH=$R
timeout -k 10 500 python -u bench/convergence.py --model vit_b_16 --image_size 224 --num_classes 10 --batch_size 64 --steps 400 --window 50 \
  --optimizer adamw --lr 3e-4 --noise 0.1 --backends native,oracle --timeout 400 > gpurun_out/r3/conv_vit.log 2>&1 || { tail -20 gpurun_out/r3/conv_vit.log; exit 1; }
tail -8 gpurun_out/r3/conv_vit.log
timeout -k 10 600 python -u bench/convergence.py --model resnet50 --image_size 224 --num_classes 100 --batch_size 64 --steps 200 --window 25 \
  --backends native,oracle --timeout 500 > gpurun_out/r3/conv_r50_224.log 2>&1 || { tail -20 gpurun_out/r3/conv_r50_224.log; exit 1; }
tail -8 gpurun_out/r3/conv_r50_224.log
