R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 400 python bench.py --offline --steps 3 --warmup 1 > $O/bench_offline.log 2> $O/bench_offline.err
timeout 400 python bench.py --ch-mode M --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_M.log 2> $O/bench_M.err
timeout 400 python bench.py --nch 2 --nb 192 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2mic.log 2> $O/bench_2mic.err
for f in offline M 2mic; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.log").read())
print("$f", d["value"], d["ms_per_step"], d["whole_path_tflops"], d["parity"], {k:(v["ms_per_step"],v["tflops"]) for k,v in d["kernels"].items() if k.startswith("lstm")})
PY
tail -2 $O/bench_$f.err; done
