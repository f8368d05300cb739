#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r34; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python tools/train_layer_bench.py 2>&1 | grep split | tee $O/layers.txt
timeout 600 python tools/train_bench.py --steps 3 2>&1 | tail -1 | tee $O/train_c4.json | cut -c1-900
