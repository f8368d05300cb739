R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; rm -f $O/nw_scan.log
for pr in 128 120 112 104 96 64 32; do
  echo "== pairs $pr (nt=256: $((pr*32)) wave-tasks)" >> $O/nw_scan.log
  timeout 200 python tools/lstm_bench.py --pairs $pr --nt 256 --layers full128s --variants 0 --reps 2 2>&1 | grep "variant" >> $O/nw_scan.log
done
cat $O/nw_scan.log
