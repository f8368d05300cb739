#!/bin/bash
# round-1 step 11: full GPU suite + IPDnet throughput
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r11
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r11/pytest.log
timeout 600 python tools/ipdnet_bench.py 2>&1 | tail -3 | tee gpurun_out/r11/ipdnet_c3.json
timeout 600 python tools/ipdnet_bench.py --mics 2 --hidden 128 --nb 64 2>&1 | tail -3 | tee gpurun_out/r11/ipdnet_2mic.json
