set -e
timeout -k 10 300 python -u bench/kernel_check.py --only conv | tail -2
timeout -k 10 280 python -u bench/resnet50_roofline.py --batch 1024 --out gpurun_out/roofline_soff.json > gpurun_out/roofline_soff.md 2>&1
grep -E "conv_fwd  |conv_dgrad  " gpurun_out/roofline_soff.md | tail -2
timeout -k 10 300 python bench.py --steps 20 --warmup 5
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --batch_size 256
