# Measurement record of the cluster-resident bf16 LSTM kernels: default bench line, kernel stats of the same command,
# L2 (TCC) request counters of the narrow-band layer alone (cluster kernel and pair-split kernel in one pass).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k; mkdir -p $O; cd $R
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
cp $(ls $O/prof/*kernel_stats.csv | head -1) $O/kernel_stats_bench_default_steps3.csv; rm -rf $O/prof
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc -o p -- python $R/tools/cluster_check.py time > $O/pmc.log 2>&1
python $R/tools/pmc_summary.py $(ls $O/pmc/*counter_collection.csv | head -1) "bf16c|bf16p" > $O/pmc_tcc_narrow_band.json; rm -rf $O/pmc
cd $R; cut -c1-300 $O/bench.json; tail -2 $O/bench.err; cat $O/pmc_tcc_narrow_band.json; head -14 $O/kernel_stats_bench_default_steps3.csv | cut -c1-150
