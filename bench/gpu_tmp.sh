set -e
timeout -k 10 300 python -u bench/kernel_check.py --only conv,stem | tail -2
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 20 --warmup 5
timeout -k 10 300 python bench.py --steps 30 --warmup 8 --batch_size 256
