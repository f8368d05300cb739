#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r28; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
FNSSL_LSTM_SPLIT=1 timeout 300 python tools/latency_bench.py 2>&1 | tail -1 | tee $O/latency_nosplit.json
timeout 300 python tools/latency_bench.py 2>&1 | tail -1 | tee $O/latency_split.json
timeout 300 python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | cut -c1-200
