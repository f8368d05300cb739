set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider 2>&1 | tail -40 > $O/pytest.log
timeout 300 python tools/lstm_bench.py --layers narrow256s,full128s,narrow256_first,full128_first --variants 0,4,5 --reps 2 > $O/lstm_bench.log 2>&1
timeout 400 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err
tail -6 $O/pytest.log; cat $O/lstm_bench.log; cat $O/bench.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print({k:d[k] for k in ('value','ms_per_step','whole_path_tflops','roofline','parity','kernels')})
"; tail -4 $O/bench.err
