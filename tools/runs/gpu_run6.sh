set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider 2>&1 | tail -40 > $O/pytest.log
timeout 200 python tools/lstm_bench.py --layers narrow256s,full128s --variants 4,5,8 --reps 2 > $O/lstm_bench.log 2>&1
tail -6 $O/pytest.log; cat $O/lstm_bench.log
