# Final lines of round 3: GPU tests, smoke, default bench (all configs), rocprofv3 kernel stats of the same command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/o; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
cp $(ls $O/prof/*kernel_stats.csv | head -1) $O/kernel_stats_bench_default_steps3.csv; rm -rf $O/prof
cd $R; cat $O/pytest.log; tail -1 $O/smoke.log; cut -c1-260 $O/bench.json; head -8 $O/kernel_stats_bench_default_steps3.csv | cut -c1-150
