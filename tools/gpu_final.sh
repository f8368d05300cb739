# Round-end measurement: GPU tests, smoke, bench (with CPU baseline), rocprofv3 kernel stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 500 python bench.py > $O/bench.log 2> $O/bench.err
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_final.log 2>&1
cd $R; tail -3 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-600 $O/bench.log; tail -3 $O/bench.err; head -8 $O/prof_final/r1_kernel_stats.csv | cut -c1-200
