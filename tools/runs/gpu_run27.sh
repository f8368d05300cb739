#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r27; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
C="SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_ipd -o p -- python $R/tools/ipdnet_bench.py --steps 1 --warmup 0 > $O/pmc_ipd.log 2>&1
timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_train -o p -- python $R/tools/train_layer_bench.py > $O/pmc_train.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o tr -- python $R/tools/train_bench.py --steps 1 --warmup 1 > $O/prof_train.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ipd -o ipd -- python $R/tools/ipdnet_bench.py --steps 2 --warmup 1 > $O/prof_ipd.log 2>&1
cd $R
python tools/pmc_summary.py $(ls $O/pmc_ipd/*counter_collection.csv | head -1) "conv3x3|lstm" > $O/pmc_ipdnet.json
python tools/pmc_summary.py $(ls $O/pmc_train/*counter_collection.csv | head -1) "lstm" > $O/pmc_train_layers.json
head -c 1500 $O/pmc_ipdnet.json; head -c 1200 $O/pmc_train_layers.json
rm -rf $O/pmc_ipd $O/pmc_train
