# Round-6 measurement record (run on the GPU box through gpurun; writes gpurun_out/final6/, copied to profiles/r06/):
# GPU tests, smoke, the default bench line (compact stdout + detail file), rocprofv3 kernel stats of the SAME command, the
# one-utterance / streaming latency table, HBM-side traffic of every roofline kernel (FETCH_SIZE and WRITE_SIZE in SEPARATE
# --pmc passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"), MFMA-busy of the config-2 kernels.
# Usage: bash tools/gpu_final_r06.sh [quick]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final6; mkdir -p $O; cd $R
if [ "$1" != "quick" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
fi
timeout 900 python bench.py --steps 20 --warmup 5 --detail gpurun_out/final6/bench_detail.json > $O/bench_default.json 2> $O/bench.err
timeout 300 python tools/latency_bench.py --json $O/latency.json > $O/latency.txt 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --ab-steps 0 --sustained-seconds 0 --detail gpurun_out/final6/prof_detail.json > $O/prof.log 2>&1
cp $(ls $O/prof/*kernel_stats.csv | head -1) $O/kernel_stats_bench_default_steps3.csv; rm -rf $O/prof
PAT="lstm_static4|lstm_static3|lstm_bf16c_kernel<256|lstm_bwd2_kernel<256|sn_mamba_scan|lstm_f32c_kernel<256|lstm_f32c_kernel<128"
pmc() {  # tag, bench arguments
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --other-configs "" --ab-steps 0 --sustained-seconds 0 --detail gpurun_out/final6/pmc_detail.json > $O/pmc_${tag}_$c.log 2>&1
    python $R/tools/pmc_summary.py $(ls $O/pmc_$c/*counter_collection.csv | head -1) "$PAT" > $O/pmc_${tag}_$c.json; rm -rf $O/pmc_$c
  done
}
pmc c2 --config 2
pmc c3 --config 3
pmc c4 --config 4
pmc c5 --config 5
pmc c2b1 --config 2 --nb 1
pmc c2s --config 2 --nb 1 --stream-chunk 12
pmc c2M --config 2 --ch-mode M
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pm -o p -- python $R/bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --other-configs "" --ab-steps 0 --sustained-seconds 0 --detail gpurun_out/final6/pmc_detail.json > $O/mfma_c2.log 2>&1
python $R/tools/pmc_summary.py $(ls $O/pm/*counter_collection.csv | head -1) "lstm_static4|lstm_f32c_kernel" > $O/pmc_mfma_busy_c2.json; rm -rf $O/pm
cd $R; python tools/hbm_traffic_r06.py $O > $O/hbm_traffic.json
cat $O/pytest.log 2>/dev/null; tail -1 $O/smoke.log 2>/dev/null; cat $O/bench_default.json; tail -3 $O/bench.err
head -14 $O/kernel_stats_bench_default_steps3.csv | cut -c1-170; cat $O/hbm_traffic.json | grep -E '^ "|ratio|bytes_per_launch"'; cat $O/pmc_mfma_busy_c2.json | grep -E "void|mfma_busy"
