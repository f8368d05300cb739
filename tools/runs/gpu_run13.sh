R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider 2>&1 | tail -8 > $O/pytest.log
echo "== STAG" > $O/lstm_bench.log
timeout 300 python tools/lstm_bench.py --layers narrow256s,narrow256_first,full128s,full128_first --variants 0 --reps 3 >> $O/lstm_bench.log 2>&1
echo "== NOSTAG" >> $O/lstm_bench.log
FNSSL_STATIC_NOSTAG=1 timeout 300 python tools/lstm_bench.py --layers narrow256s,narrow256_first,full128s,full128_first --variants 0 --reps 3 >> $O/lstm_bench.log 2>&1
tail -5 $O/pytest.log; grep -v amdgpu.ids $O/lstm_bench.log
