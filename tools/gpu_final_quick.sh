# Quick end-of-round record (run on the GPU box through gpurun; writes gpurun_out/final6/, copied to profiles/r05/k_* and n_*):
# all GPU tests, smoke(), the default bench line at 20 steps, the one-utterance / streaming latency table.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final6; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err
cat $O/pytest.log; tail -1 $O/smoke.log; cut -c1-260 $O/bench_default.json; tail -2 $O/bench.err
mkdir -p $O; timeout 300 python tools/latency_bench.py --json $O/latency.json > $O/latency.txt 2>&1; grep -E "^utt|^batch|^chunk" $O/latency.txt | cut -c1-200
