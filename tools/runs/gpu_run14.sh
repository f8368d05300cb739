R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; rm -f $O/xd8.log
for x in 0 1; do
 for pr in 128 96 64; do
  echo "== XD8=$x pairs $pr" >> $O/xd8.log
  FNSSL_STATIC_XD8=$x timeout 200 python tools/lstm_bench.py --pairs $pr --nt 256 --layers full128s --variants 0 --reps 2 2>&1 | grep "variant" >> $O/xd8.log
 done
 echo "== XD8=$x full config" >> $O/xd8.log
 FNSSL_STATIC_XD8=$x timeout 200 python tools/lstm_bench.py --layers full128s --variants 0 --reps 2 2>&1 | grep "variant" >> $O/xd8.log
done
cat $O/xd8.log
