#!/bin/bash
# Samples the card's shader clock / power while (a) the step replays, (b) one GEMM shape is re-launched back to back.
# usage (on the GPU box): bash tools/clock_watch.sh > gpurun_out/clock_watch.log
cd "$(dirname "$0")/.."
sample() {      # $1 = label, $2 = pid to watch
  for i in 1 2 3 4 5 6 7 8; do
    sleep 2
    kill -0 "$2" 2>/dev/null || break
    echo "--- $1 sample $i"
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" 
  done
}
echo "=== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power"
python bench.py --steps 2500 --warmup 8 --no-cpu-baseline --no-kernel-timing > gpurun_out/clock_bench.log 2>&1 &
P=$!
sleep 25
sample step $P
wait $P
tail -1 gpurun_out/clock_bench.log | cut -c1-200
python tools/gemm_hold.py 22 > gpurun_out/clock_gemm.log 2>&1 &
P=$!
sleep 12
sample gemm $P
wait $P
