#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r29; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python bench.py --steps 5 --no-cpu-baseline 2>/dev/null | tee $O/bench_peel.json | cut -c1-260
FNSSL_LSTM_NO_PEEL=1 timeout 300 python bench.py --steps 5 --no-cpu-baseline 2>/dev/null | tee $O/bench_nopeel.json | cut -c1-260
python - <<'PY'
import json
for f in ("bench_peel","bench_nopeel"):
    d=json.loads(open("gpurun_out/r29/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], {k:(round(v["ms"],1), v.get("tflops")) for k,v in d["kernels"].items() if "lstm" in k})
PY
