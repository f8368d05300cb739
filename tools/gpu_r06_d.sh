# Round 6: the four-slice operand-ring kernel (lstm_static4.h) — bit-identity tests, then config 2 with the A/B leg.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "operand_ring or config2_full or host_logic" 2>&1 | tail -15 > $O/pytest_static4.log
cat $O/pytest_static4.log | tail -6
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 --ab-steps 4 --other-configs "" --detail gpurun_out/r06d/c2_detail.json > $O/c2.json 2> $O/c2.err
cat $O/c2.json; grep "A/B\|timed\|parity" $O/c2.err
