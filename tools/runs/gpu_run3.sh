set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider 2>&1 | tail -30 > $O/pytest.log
timeout 400 python tools/lstm_bench.py --layers narrow256s,full128s,narrow256_first,full128_first > $O/lstm_bench.log 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/prof_pmc1 -o p1 -- python $R/tools/lstm_bench.py --layers narrow256s,full128s --variants 4,5,8 --reps 1 > $O/prof_pmc1.log 2>&1
cd $R
timeout 500 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err
tail -4 $O/pytest.log; cat $O/lstm_bench.log; cat $O/bench.log; tail -8 $O/bench.err
