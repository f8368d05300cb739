#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r32; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bf16" 2>&1 | tail -2 | tee $O/pytest.log
timeout 500 python bench.py --bf16 --steps 3 --cpu-seconds 10 2> $O/bench_bf16.err | tee $O/bench_bf16.json | cut -c1-400
tail -3 $O/bench_bf16.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r32/bench_bf16.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["parity"], d["roofline"]); print(d["kernels"])
PY
