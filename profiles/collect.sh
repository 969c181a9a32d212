#!/bin/bash
# Collects the measurements behind bench.py's roofline/traffic fields on the GPU box.
#   usage (repo root, MI355X box):  bash profiles/collect.sh r01_d
# Writes gpurun_out/<tag>/...; profiles/summarize.py turns that into the committed summaries under profiles/.
set -u
tag=${1:-r01_x}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
python3 bench.py --steps 50 --warmup 10 > "$out/bench_4096.json" 2> "$out/bench_4096.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs-per-gpu 65536 > "$out/bench_65536.json" 2> "$out/bench_65536.err"
python3 tests/bench_configs.py > "$out/other_configs.jsonl" 2> /dev/null
python3 tests/bench_cassie3d.py --cpu-baseline >> "$out/other_configs.jsonl" 2> /dev/null
python3 train_trpo.py --envs-per-gpu 65536 --horizon 8 --n-itr 5 --timing > "$out/trpo_65536.jsonl" 2> /dev/null
cd /tmp && export TMPDIR=/tmp
# per-kernel durations of the same command as the bench line (kernel trace + stats only)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 "$root/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > "$out/stats_bench.log" 2>&1
# HBM traffic counters, one counter per pass, no tracing domains
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -o pmc -- python3 "$root/tests/prof_step.py" 4096 6 > "$out/pmc_$c.log" 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc_osc_$c" -o pmc -- python3 "$root/tests/prof_step.py" 4096 6 OSC > "$out/pmc_osc_$c.log" 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc_3d_$c" -o pmc -- python3 "$root/tests/bench_cassie3d.py" --envs 4096 --steps 6 > "$out/pmc_3d_$c.log" 2>&1
done
ls -R "$out" | head -50
