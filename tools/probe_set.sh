set -e
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -x -k "gru" 2>&1 | tail -3
timeout -k 10 300 python bench.py --config large --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null
