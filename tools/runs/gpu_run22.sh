#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r22; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
timeout 600 python tools/train_bench.py --steps 2 2>&1 | tail -1 | tee $O/train_c4.json | cut -c1-1500
FNSSL_TRAIN_SPLIT=2 timeout 600 python tools/train_bench.py --steps 2 2>&1 | tail -1 | tee $O/train_c4_split2.json | cut -c1-1500
FNSSL_TRAIN_SPLIT=1 timeout 600 python tools/train_bench.py --steps 2 2>&1 | tail -1 | tee $O/train_c4_split1.json | cut -c1-900
