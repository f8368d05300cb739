# H = 256 cluster kernel at the 'M'-pairing size (96 pairs = 1536 groups = 96 per cluster): how far the members may drift.
# HISTORICAL: the F32C_DRIFT knob this script drove was removed after the measurement (profiles/r05/c_*: no effect); to repeat it,
# re-instantiate launch_f32c_k<256, 16, 0, kSum, false, DRIFT, 8> in csrc/lstm_f32c.hip for the values of interest.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
for d in 2 1 3 4 8; do
  FNSSL_F32C_DRIFT=$d python bench.py --ch-mode M --steps 4 --warmup 2 --no-cpu-baseline --other-configs "" --ab-steps 0 > $O/drift_$d.json 2> $O/drift_$d.err
  python -c "
import json; d=json.load(open('$O/drift_$d.json')); print('drift', $d, d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'], d['cluster_fallbacks'])"
done
