R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; rm -f $O/ablate.log
for a in 0 1 2 4 8 16 32 5 13 63; do
  echo "== ABLATE $a" >> $O/ablate.log
  FNSSL_ABLATE=$a timeout 200 python tools/lstm_bench.py --layers narrow256s,full128s --variants 0 --reps 2 2>&1 | grep -E "variant 0" >> $O/ablate.log
done
cat $O/ablate.log
