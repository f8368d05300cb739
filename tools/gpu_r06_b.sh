# Round 6, second run: the bf16 cluster split (H = 128 full-band layers as clusters of 17 - 20 tiles), PD localisation tests,
# config 3 alone with its A/B leg, config 4 with and without pair chunks (memory).  Usage (GPU box): bash tools/gpu_r06_b.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "bf16 or doa or cluster" 2>&1 | tail -15 > $O/pytest_bf16.log
timeout 300 python bench.py --config 3 --steps 20 --warmup 3 --ab-steps 5 --other-configs "" --detail gpurun_out/r06b/c3_detail.json > $O/c3.json 2> $O/c3.err
for cp in 0 16 8; do
  timeout 300 python bench.py --config 4 --steps 4 --warmup 2 --chunk-pairs $cp --no-cpu-baseline --other-configs "" --sustained-seconds 0 --detail gpurun_out/r06b/c4_chunk$cp.json > $O/c4_chunk$cp.line.json 2> $O/c4_chunk$cp.err
done
cat $O/pytest_bf16.log | tail -5; cat $O/c3.json; tail -5 $O/c3.err
for cp in 0 16 8; do python - <<PY
import json
d=json.load(open("$O/c4_chunk$cp.line.json")); print("chunk $cp:", d["ms_per_step"], "ms", d["peak_mem_gb"], "GB", d["value"])
PY
done
