#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r24; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.log
for s in 0 4; do
FNSSL_TRAIN_SPLIT=$s timeout 600 python tools/train_bench.py --steps 2 2>&1 | tail -1 | tee $O/train_c4_split$s.json | cut -c1-1000
done
