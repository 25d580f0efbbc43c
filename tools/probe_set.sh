for abl in 1 3 4; do
echo "abl $abl"
CPC_NCE_ABL=$abl timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null
done
