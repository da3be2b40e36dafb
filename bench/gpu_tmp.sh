set -e
timeout -k 10 300 python -u bench/kernel_check.py | tail -2
PDT_WGRAD_WIN=1 timeout -k 10 300 python -u bench/kernel_check.py --only conv | tail -2
timeout -k 10 300 python bench.py --steps 20 --warmup 5
PDT_WGRAD_WIN=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5
