#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r25; mkdir -p $O; cd $R
for s in 1 2 4; do FNSSL_TRAIN_SPLIT=$s timeout 300 python tools/train_layer_bench.py 2>&1 | grep split | tee -a $O/layers.txt; done
