set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for a in 0 63 15 1; do
FNSSL_ABLATE=$a timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/prof_abl$a -o p -- python $R/tools/lstm_bench.py --layers narrow256s,full128s --variants 4,5 --reps 1 > $O/prof_abl$a.log 2>&1
done
ls $O/prof_abl0
