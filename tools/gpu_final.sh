# Round-end measurement: GPU tests, smoke, bench (with CPU baseline), rocprofv3 kernel stats of the same command.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
cp $(ls $O/prof/*kernel_stats.csv | head -1) $O/kernel_stats_bench_steps3.csv; rm -rf $O/prof
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc.log 2>&1
python $R/tools/pmc_summary.py $(ls $O/pmc/*counter_collection.csv | head -1) "lstm" > $O/pmc_lstm_kernels.json; rm -rf $O/pmc
cd $R; tail -3 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-500 $O/bench.json; tail -3 $O/bench.err; head -8 $O/kernel_stats_bench_steps3.csv | cut -c1-160
