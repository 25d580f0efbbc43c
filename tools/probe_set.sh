CPC_NCE_ABL=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null
