# SQ counter passes of the cluster kernel alone (narrow-band layer, config 3's batch): where do the waves wait
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pm; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  NBS=${NBS:-64} LAYER=${LAYER:-narrow} timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python $R/tools/cluster_check.py time > $O/p$i.log 2>&1
  python $R/tools/pmc_summary.py $(ls $O/p$i/*counter_collection.csv | head -1) "bf16c" > $O/pmc_set$i.json; rm -rf $O/p$i
  cat $O/pmc_set$i.json
done
