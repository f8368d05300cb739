#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for s4 in 6 8 10 12; do for nb in 20 24 28 32; do
  FNSSL_SPLIT4_MAX_H256=$s4 timeout 200 python bench.py --nb $nb --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('s4max=$s4 nb=%-3d %9.1f frames/s %8.2f ms  h128 %.1f ms  h256 %.1f ms' % ($nb, d['value'], d['ms_per_step'], k['lstm_h128']['ms_per_step'], k['lstm_h256']['ms_per_step']))"
done; done
