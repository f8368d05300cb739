#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r19
timeout 600 python -m pytest tests -m gpu -x -q -k "static or ipdnet" 2>&1 | tail -5 | tee gpurun_out/r19/pytest.log
timeout 300 python tools/ipdnet_bench.py 2>&1 | tail -1 | tee gpurun_out/r19/ipdnet_c3.json
FNSSL_NO_STATIC_IPDNET=1 timeout 300 python tools/ipdnet_bench.py 2>&1 | tail -1 | cut -c1-900 | tee gpurun_out/r19/ipdnet_c3_generic.json
