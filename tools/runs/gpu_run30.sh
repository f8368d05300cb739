#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r30; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bf16" 2>&1 | grep -E "^E  |passed|failed|^FAILED|Error" | cut -c1-250 | head -30 | tee $O/pytest_bf16.log
timeout 300 python tools/ipdnet_bench.py --bf16 2>&1 | tail -1 | tee $O/ipdnet_c3_bf16.json | cut -c1-1200
