#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r26; mkdir -p $O; cd $R
for s in 2 4; do FNSSL_TRAIN_SPLIT=$s timeout 300 python tools/train_layer_bench.py 2>&1 | grep split | tee -a $O/layers.txt; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.log
timeout 600 python tools/train_bench.py --steps 2 2>&1 | tail -1 | tee $O/train_c4.json | cut -c1-1000
