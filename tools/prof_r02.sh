# rocprofv3 evidence for the default bench command (run ON the GPU box: gpurun -- 'bash tools/prof_r02.sh [tag]'):
#   kernel-trace stats, separate FETCH_SIZE / WRITE_SIZE PMC passes -> per-kernel HBM traffic (stamped with the library hash),
#   the default bench line and the per-layer table.  Only the small summaries are kept (gpurun_out/<tag>_keep/).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
W=gpurun_out/${TAG}_work; K=gpurun_out/${TAG}_keep
mkdir -p $W $K
python3 bench.py > $K/${TAG}_bench_cfg2_bf16.json 2> $W/bench.err; echo "bench rc=$?"; cut -c1-400 $K/${TAG}_bench_cfg2_bf16.json
python3 bench.py --no-cpu-baseline --per-layer > $W/pl.json 2>> $W/bench.err; python3 tools/per_layer_table.py $W/pl.json > $K/${TAG}_per_layer_cfg2.txt; echo "per-layer rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $W/bench_stats.log 2>&1
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $W/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 0 > $W/bench_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $W/pmc_write -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 0 > $W/bench_write.log 2>&1
echo "write rc=$?"
python3 tools/pmc_summary.py $W/pmc_fetch $W/pmc_write $K/pmc_traffic_cfg2_bf16.json | head -8
f=$(find $W/stats -name "*kernel_stats.csv" | head -1); cp "$f" $K/${TAG}_bench_cfg2_kernel_stats.csv; head -6 "$f" | cut -c1-160
tail -1 $W/bench_stats.log | cut -c1-300 > $K/${TAG}_bench_under_rocprof.txt
rm -rf $W
