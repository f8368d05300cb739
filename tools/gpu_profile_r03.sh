# Round-3 measurement record (run on the GPU box through gpurun; writes gpurun_out/final3/, copied to profiles/r03/f_*):
# default bench line, rocprofv3 kernel stats of the SAME command, HBM-side traffic of the H = 256 LSTM kernel (FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"), SQ counters of the training kernels.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final3; mkdir -p $O; cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
cp $(ls $O/prof/*kernel_stats.csv | head -1) $O/kernel_stats_bench_default_steps3.csv; rm -rf $O/prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --other-configs "" > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_summary.py $(ls $O/pmc_$c/*counter_collection.csv | head -1) "lstm" > $O/pmc_c2_$c.json; rm -rf $O/pmc_$c
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc4 -o p -- python $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc4.log 2>&1
python $R/tools/pmc_summary.py $(ls $O/pmc4/*counter_collection.csv | head -1) "wgrad|lstm_bwd|lstm_split|combine" > $O/pmc_c4_sq.json; rm -rf $O/pmc4
cd $R; cut -c1-300 $O/bench.json; tail -2 $O/bench.err; head -12 $O/kernel_stats_bench_default_steps3.csv | cut -c1-160; cat $O/pmc_c2_FETCH_SIZE.json | head -30
