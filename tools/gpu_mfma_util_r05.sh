# MFMA-pipe utilisation of the fp32 LSTM kernels from counters (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE; its own
# --pmc pass per shape): config 2 (lstm_static3 / lstm_f32c<128>), the 'M' pairing (lstm_f32c<256> at 96 groups per cluster),
# one 4-mic utterance and a streaming chunk.  Usage (GPU box): bash tools/gpu_mfma_util_r05.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mfma5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PAT="lstm_static3|lstm_f32c_kernel"
run() {  # tag, bench arguments
  tag=$1; shift
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/p -o p -- python $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --other-configs "" --ab-steps 0 > $O/$tag.log 2>&1
  python $R/tools/pmc_summary.py $(ls $O/p/*counter_collection.csv | head -1) "$PAT" > $O/mfma_$tag.json; rm -rf $O/p
  python - <<PY
import json
d = json.load(open("$O/mfma_$tag.json"))
for k, v in d.items():
    print("$tag", k[:100], "launches", v.get("launches"), "mfma_busy_frac", v.get("mfma_busy_frac"))
PY
}
run c2 --config 2
run c2M --config 2 --ch-mode M
run c2b1 --config 2 --nb 1
run c2s --config 2 --nb 1 --stream-chunk 12
