# Round-6 first measurement: GPU tests, smoke, the default bench line (compact stdout + detail file), and the counter passes
# for config 3's bf16 kernels the round-5 review asked for (MFMA busy, VALU / VMEM / LDS issue, LDS bank conflicts: separate
# --pmc passes, no traces beside them).   Usage (GPU box): bash tools/gpu_r06_a.sh [quick]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
if [ "$1" != "quick" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > $O/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
fi
timeout 900 python bench.py --steps 20 --warmup 5 --detail gpurun_out/r06a/bench_detail.json > $O/bench_default.json 2> $O/bench.err
wc -c $O/bench_default.json
export TMPDIR=/tmp; cd /tmp
PAT="lstm_bf16c_kernel|conv3x3_bf16|lstm_bf16"
pass() {  # tag, counters...
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/p -o p -- python $R/bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline --other-configs "" --ab-steps 0 --sustained-seconds 0 > $O/pmc_c3_$tag.log 2>&1
  python $R/tools/pmc_summary.py $(ls $O/p/*counter_collection.csv | head -1) "$PAT" > $O/pmc_c3_$tag.json; rm -rf $O/p
}
pass mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pass sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA
cd $R
cat $O/pytest.log 2>/dev/null | tail -4; tail -1 $O/smoke.log 2>/dev/null; cat $O/bench_default.json; tail -12 $O/bench.err
for t in mfma sq lds; do echo "== $t"; cat $O/pmc_c3_$t.json | head -60; done
