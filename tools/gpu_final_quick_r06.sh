# HEAD check at the end of round 6: GPU tests, smoke, the default bench line (compact + detail).  Usage: bash tools/gpu_final_quick_r06.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final6b; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --detail gpurun_out/final6b/bench_detail.json > $O/bench_default.json 2> $O/bench.err
cat $O/pytest.log; tail -1 $O/smoke.log; wc -c $O/bench_default.json; cat $O/bench_default.json; tail -3 $O/bench.err
