#!/bin/bash
# IPDnet config-3 geometry: rocprofv3 kernel stats + planner info
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r18; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ipd -- python $R/tools/ipdnet_bench.py --steps 2 --warmup 1 > $O/prof.log 2>&1
cd $R; tail -2 $O/prof.log | cut -c1-400; head -12 $O/prof/ipd_kernel_stats.csv | cut -c1-260
