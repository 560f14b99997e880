# rocprofv3 evidence for the default bench command (run ON the GPU box: gpurun -- 'bash tools/prof_r04.sh [tag]'):
#   1. separate FETCH_SIZE / WRITE_SIZE PMC passes -> per-kernel HBM traffic, stamped with the sha256 of the library in use
#      (written to profiles/ of the box's copy too, so that step 3 reports roofline.traffic from THIS build),
#   2. kernel-trace stats of the same command,
#   3. the default bench line (with the CPU baseline, the event-free region and the fp8 mode) and the per-layer tables.
# Only the small summaries are kept (gpurun_out/<tag>_keep/); copy them into profiles/ afterwards.
TAG=${1:-r04e}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
W=gpurun_out/${TAG}_work; K=gpurun_out/${TAG}_keep
mkdir -p $W $K
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $W/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-kernel-timing --no-extras --steps 5 --warmup 0 > $W/bench_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $W/pmc_write -- python3 bench.py --no-cpu-baseline --no-kernel-timing --no-extras --steps 5 --warmup 0 > $W/bench_write.log 2>&1
echo "write rc=$?"
python3 tools/pmc_summary.py $W/pmc_fetch $W/pmc_write $K/pmc_traffic_cfg2_bf16.json | head -8
cp $K/pmc_traffic_cfg2_bf16.json profiles/pmc_traffic_cfg2_bf16.json
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 bench.py --no-cpu-baseline --no-kernel-timing --no-extras --steps 20 --warmup 5 > $W/bench_stats.log 2>&1
echo "stats rc=$?"
f=$(find $W/stats -name "*kernel_stats.csv" | head -1); cp "$f" $K/${TAG}_bench_cfg2_kernel_stats.csv; head -6 "$f" | cut -c1-160
grep '^{"metric"' $W/bench_stats.log | cut -c1-400 > $K/${TAG}_bench_under_rocprof.json
python3 bench.py > $K/${TAG}_bench_cfg2_bf16.json 2> $W/bench.err; echo "bench rc=$?"; cut -c1-300 $K/${TAG}_bench_cfg2_bf16.json
python3 bench.py --no-cpu-baseline --per-layer > $W/pl.json 2>> $W/bench.err; python3 tools/per_layer_table.py $W/pl.json > $K/${TAG}_per_layer_cfg2.txt; echo "per-layer rc=$?"
python3 bench.py --dtype fp8 --no-cpu-baseline --per-layer > $K/${TAG}_bench_cfg2_fp8_per_layer.json 2>> $W/bench.err; python3 tools/per_layer_table.py $K/${TAG}_bench_cfg2_fp8_per_layer.json > $K/${TAG}_per_layer_cfg2_fp8.txt; sed -n 2p $K/${TAG}_per_layer_cfg2_fp8.txt | cut -c1-80
rm -rf $W
