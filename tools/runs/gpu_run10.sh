R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace2 -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_trace2.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o p -- python $R/tools/lstm_bench.py --layers narrow256s,full128s --variants 0 --reps 1 > $O/prof_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o p -- python $R/tools/lstm_bench.py --layers narrow256s,full128s --variants 0 --reps 1 > $O/prof_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/prof_pmc3 -o p -- python $R/tools/lstm_bench.py --layers narrow256s,full128s --variants 0 --reps 1 > $O/prof_pmc3.log 2>&1
cd $R; timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench.log 2> $O/bench.err; cat $O/bench.log | cut -c1-400; tail -3 $O/bench.err
