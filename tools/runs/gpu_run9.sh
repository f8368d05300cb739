R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider 2>&1 | tail -5 > $O/pytest.log
echo "== default (mid-chunk staging, big chunks for H=128)" > $O/lstm_bench.log
timeout 300 python tools/lstm_bench.py --layers narrow256s,full128s,narrow256_first,full128_first --variants 0 --reps 2 >> $O/lstm_bench.log 2>&1
echo "== FNSSL_STATIC_SMALLCHUNK=1" >> $O/lstm_bench.log
FNSSL_STATIC_SMALLCHUNK=1 timeout 300 python tools/lstm_bench.py --layers full128s,full128_first --variants 0 --reps 2 >> $O/lstm_bench.log 2>&1
tail -3 $O/pytest.log; cat $O/lstm_bench.log
