#!/bin/bash
# throughput vs batch size (utterances per GPU), FN-SSL fp32, 4 mics, 300 frames
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r35; mkdir -p $O; cd $R
for nb in 1 2 4 8 16 32 64; do
  timeout 200 python bench.py --nb $nb --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nb=%-3d pairs=%-4d %9.1f frames/s  %8.2f ms/step  %6.1f TF/s' % ($nb, $nb*6, d['value'], d['ms_per_step'], d['whole_path_tflops']))" | tee -a $O/batch_scan.txt
done
