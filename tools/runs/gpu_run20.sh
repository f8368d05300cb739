#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r20
timeout 900 python tools/train_bench.py --steps 2 2>&1 | tail -2 | tee gpurun_out/r20/train_c4.json
