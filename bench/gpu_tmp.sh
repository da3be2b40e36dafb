set -e
timeout -k 10 300 python -u bench/kernel_check.py --only conv | tail -3
timeout -k 10 280 python -u bench/resnet50_roofline.py --batch 1024 --out gpurun_out/roofline_wg2.json > gpurun_out/roofline_wg2.md 2>&1
grep -E "conv_wgrad  |conv_fwd  |conv_dgrad  " gpurun_out/roofline_wg2.md | tail -3
timeout -k 10 300 python bench.py --steps 20 --warmup 5
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --batch_size 256
