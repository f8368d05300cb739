R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; rm -f $O/nw_scan.log
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
for pr in 128 104 96 64 32; do
  echo "== full128s pairs $pr (nt=256)" >> $O/nw_scan.log
  timeout 200 python tools/lstm_bench.py --pairs $pr --nt 256 --layers full128s,full128_first --variants 0 --reps 2 2>&1 | grep "variant" >> $O/nw_scan.log
done
for pr in 192 128 64; do
  echo "== narrow256s pairs $pr (nt=300)" >> $O/nw_scan.log
  timeout 200 python tools/lstm_bench.py --pairs $pr --layers narrow256s,narrow256_first --variants 0 --reps 2 2>&1 | grep "variant" >> $O/nw_scan.log
done
tail -3 $O/pytest.log; cat $O/nw_scan.log
