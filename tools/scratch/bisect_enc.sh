ROOT=$PWD
for sha in 35d0e61 a9f6927 HEAD; do
  if [ $sha = HEAD ]; then d=$ROOT; else d=$ROOT/tools/scratch/wt_$sha; fi
  cd $d
  for n in 16 40; do echo "== $sha n=$n"; python tools/scratch/enc_err.py $n 2>&1 | grep -E "conv0.weight|conv1.weight|conv3.weight|conv4.weight"; done
done
