cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02f
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02f/stats -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > gpurun_out/r02f/bench_stats.log 2>&1
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02f/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 0 > gpurun_out/r02f/bench_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02f/pmc_write -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 0 > gpurun_out/r02f/bench_write.log 2>&1
echo "write rc=$?"
python3 tools/pmc_summary.py gpurun_out/r02f/pmc_fetch gpurun_out/r02f/pmc_write gpurun_out/r02f/pmc_traffic_cfg2_bf16.json | head -14
find gpurun_out/r02f/stats -name "*kernel_stats.csv" | head -2
f=$(find gpurun_out/r02f/stats -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-200
# keep only the small summaries (the traces are hundreds of MB)
mkdir -p gpurun_out/r02f_keep; cp "$f" gpurun_out/r02f_keep/r02f_bench_cfg2_kernel_stats.csv; cp gpurun_out/r02f/pmc_traffic_cfg2_bf16.json gpurun_out/r02f_keep/; cp gpurun_out/r02f/bench_stats.log gpurun_out/r02f_keep/
rm -rf gpurun_out/r02f
