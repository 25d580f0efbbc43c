set -e
for np in 0 1; do
echo "nopipe $np"
if [ $np = 1 ]; then export CPC_X6_NOPIPE=1; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null
done
