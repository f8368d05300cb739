set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider 2>&1 | tail -15 > $O/pytest.log
for a in 0 1 2 4 8 16 32 6 7 15 63; do
  echo "== ABLATE $a" >> $O/ablate.log
  FNSSL_ABLATE=$a timeout 200 python tools/lstm_bench.py --layers narrow256s,full128s --variants 4,5 --reps 2 2>&1 | grep -E "narrow256s +variant 4|full128s +variant 5" >> $O/ablate.log
done
tail -4 $O/pytest.log; cat $O/ablate.log
