set -e
timeout -k 10 300 python -u bench/kernel_check.py --only conv | tail -2
timeout -k 10 300 python bench.py --steps 20 --warmup 5
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --batch_size 256
