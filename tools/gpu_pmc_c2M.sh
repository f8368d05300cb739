# Fabric-side traffic of the streamed-row H = 256 cluster kernel at the 'M'-pairing size (two separate --pmc passes) -> gpurun_out/pmc2M/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2M; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --config 2 --ch-mode M --steps 1 --warmup 1 --no-cpu-baseline --other-configs "" --ab-steps 0 > $O/pmc_c2M_$c.log 2>&1
  python $R/tools/pmc_summary.py $(ls $O/pmc_$c/*counter_collection.csv | head -1) "lstm_f32c_kernel<256" > $O/pmc_c2M_$c.json; rm -rf $O/pmc_$c
done
cd $R; python tools/hbm_traffic_r05.py $O
