set -e
timeout -k 10 200 python -u bench/kernel_check.py --only conv | tail -2
timeout -k 10 280 python -u bench/resnet50_roofline.py --batch 1024 --out gpurun_out/roofline_km32b.json > gpurun_out/roofline_km32b.md 2>&1
grep -E "conv_wgrad  " gpurun_out/roofline_km32b.md | tail -1
timeout -k 10 300 python bench.py --steps 20 --warmup 5
