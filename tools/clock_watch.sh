# Samples the GPU's clocks and power (rocm-smi, read-only) while the default bench workload runs: what the chip actually holds under
# the step's MFMA load.  Run ON the GPU box: gpurun -- 'bash tools/clock_watch.sh [extra bench flags]'  -> gpurun_out/clock_watch.log
mkdir -p gpurun_out
( for i in $(seq 1 120); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/clock_watch.log 2>&1 &
W=$!
python3 bench.py --no-cpu-baseline --no-extras --no-kernel-timing --steps 40 --warmup 5 "$@" > gpurun_out/clock_watch_bench.json 2> gpurun_out/clock_watch_bench.err
kill $W 2>/dev/null
wait $W 2>/dev/null
cut -c1-200 gpurun_out/clock_watch_bench.json
