# Round-end measurement: GPU tests, smoke, bench (with CPU baseline), rocprofv3 kernel stats of the same command,
# then the config-5 (IPDnet2, bf16 / fp32) lines, kernel stats and counters.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --config 5 --steps 20 --warmup 5 > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err
timeout 300 python bench.py --config 5 --fp32 --steps 20 --warmup 5 > $O/bench_c5_fp32.json 2> $O/bench_c5_fp32.err
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
cp $(ls $O/prof/*kernel_stats.csv | head -1) $O/kernel_stats_bench_steps3.csv; rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5 -o r5 -- python $R/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof5.log 2>&1
cp $(ls $O/prof5/*kernel_stats.csv | head -1) $O/kernel_stats_bench_c5_bf16.csv; rm -rf $O/prof5
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc.log 2>&1
python $R/tools/pmc_summary.py $(ls $O/pmc/*counter_collection.csv | head -1) "lstm" > $O/pmc_lstm_kernels.json; rm -rf $O/pmc
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc5 -o p -- python $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc5.log 2>&1
python $R/tools/pmc_summary.py $(ls $O/pmc5/*counter_collection.csv | head -1) "sn_" > $O/pmc_c5_bf16_passA.json; rm -rf $O/pmc5
cd $R; tail -3 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-400 $O/bench.json; tail -2 $O/bench.err; cut -c1-300 $O/bench_c5_bf16.json; cut -c1-300 $O/bench_c5_fp32.json
