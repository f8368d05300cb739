#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r21; mkdir -p $O; cd $R
timeout 900 python tools/train_bench.py --steps 2 --mics 4 --chunk-pairs 48 2>&1 | tail -1 | tee $O/train_4mic.json | cut -c1-1800
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tr -- python $R/tools/train_bench.py --steps 1 --warmup 1 > $O/prof.log 2>&1
cd $R; head -24 $O/prof/tr_kernel_stats.csv | cut -c1-230
