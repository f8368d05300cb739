# rocprofv3 kernel stats of the default bench command (3 steps) for the build in the tree -> gpurun_out/final6/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final6; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --ab-steps 0 > $O/prof_bench.json 2> $O/prof.log
cp $(ls $O/prof/*kernel_stats.csv | head -1) $O/kernel_stats_bench_default_steps3.csv; rm -rf $O/prof
head -12 $O/kernel_stats_bench_default_steps3.csv | cut -c1-200; cut -c1-200 $O/prof_bench.json
