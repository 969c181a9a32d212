#!/bin/bash
# Shader clock and power while the headline kernel runs (not a test): bench.py in the background, rocm-smi polled beside it.
cd "$(dirname "$0")/.."
python3 bench.py --steps 20000 --warmup 50 --no-extra --no-cpu-baseline > /tmp/clock_probe_bench.json 2>/dev/null &
pid=$!
sleep 18
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power\|fclk" | tr -s " " | head -6
  echo ---
  sleep 0.7
done
wait $pid
tail -1 /tmp/clock_probe_bench.json | cut -c1-200
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr -s " " | head -3
