#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r23; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
timeout 600 python tools/train_bench.py --steps 2 2>&1 | tail -1 | tee $O/train_c4.json | cut -c1-1300
FNSSL_TRAIN_SPLIT=4 timeout 600 python tools/train_bench.py --steps 2 2>&1 | tail -1 | tee $O/train_c4_split4.json | cut -c1-1300
timeout 900 python tools/train_bench.py --steps 2 --mics 4 --chunk-pairs 48 2>&1 | tail -1 | tee $O/train_4mic.json | cut -c1-1300
