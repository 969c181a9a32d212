#!/bin/bash
# One GPU-box session: the -m gpu suite, the bench line as the driver runs it, and the rocprofv3 kernel statistics of the same
# command.   usage (repo root, MI355X box):  bash profiles/run_gpu.sh <tag> [tests|bench|prof ...]
set -u
tag=${1:-r02_x}; shift || true
what=${*:-tests bench prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
rm -f "$root/gpurun_out/fullsize.jsonl"
for w in $what; do
  case $w in
    tests)
      timeout 3000 python3 -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
      tail -5 "$out/pytest_gpu.log"
      [ -f "$root/gpurun_out/fullsize.jsonl" ] && cp "$root/gpurun_out/fullsize.jsonl" "$out/fullsize.jsonl" ;;
    bench)
      timeout 900 python3 bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
      tail -c 3000 "$out/bench.json" ;;
    prof)
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
          python3 "$root/bench.py" --no-extra --no-cpu-baseline > "$out/stats_bench.log" 2>&1 )
      find "$out/stats" -name "*kernel_stats.csv" | head -1 | xargs -r head -12 ;;
    legforced)   # the whole -m gpu suite once more with the two-lanes-per-environment tier forced on for every batch size
      CASSIE2D_LEG=1 timeout 3000 python3 -m pytest tests -m gpu -q > "$out/pytest_gpu_leg_forced.log" 2>&1; echo "pytest(leg forced) rc=$?" >> "$out/pytest_gpu_leg_forced.log"
      tail -3 "$out/pytest_gpu_leg_forced.log" ;;
    profosc)   # kernel statistics of configs[2]: controller kernel + physics kernel per substep (tools/prof_step.py, OSC in the loop)
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_osc" -o osc -- \
          python3 "$root/tools/prof_step.py" 65536 30 OSC > "$out/stats_osc.log" 2>&1 )
      find "$out/stats_osc" -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-170 ;;
    proffallen)   # kernel trace of the all-fallen floor (lower tiers side by side on two streams)
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_fallen" -o fallen -- \
          python3 "$root/tools/prof_fallen.py" > "$out/stats_fallen.log" 2>&1 )
      find "$out/stats_fallen" -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-170 ;;
    trpo)
      timeout 900 python3 train_trpo.py --envs-per-gpu 65536 --horizon 8 --n-itr 5 --kind stand --control-mode Torque --timing > "$out/trpo_65536.jsonl" 2> "$out/trpo.err"
      tail -2 "$out/trpo_65536.jsonl" | cut -c1-400
      # the loop's headline configuration: walk env / PD (what bench.py steps), 65 536 envs x 8 steps per iteration
      timeout 900 python3 train_trpo.py --envs-per-gpu 65536 --horizon 8 --n-itr 10 --kind walk --control-mode PD --timing > "$out/trpo_walk_65536.jsonl" 2>> "$out/trpo.err"
      tail -2 "$out/trpo_walk_65536.jsonl" | cut -c1-400 ;;
    proftrpo)   # kernel statistics of the TRPO loop (walk env / PD): Env.step, policy step, sampler step, Fisher-vector products, line search
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_trpo" -o trpo -- \
          python3 "$root/train_trpo.py" --envs-per-gpu 65536 --horizon 8 --n-itr 8 --kind walk --control-mode PD > "$out/stats_trpo.log" 2>&1 )
      find "$out/stats_trpo" -name "*kernel_stats.csv" | head -1 | xargs -r head -14 | cut -c1-170 ;;
    prof3d)   # kernel statistics of configs[4] (Cassie3d, 16 384 envs, reset every 40 steps)
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_3d" -o c3d -- \
          python3 "$root/tools/bench_cassie3d.py" > "$out/stats_3d.log" 2>&1 )
      grep -a "^{" "$out/stats_3d.log" | tail -1 > "$out/cassie3d_bench.json"
      find "$out/stats_3d" -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-170 ;;
    pmc)
      bash profiles/collect_pmc.sh $tag | tail -12
      # refresh profiles/pmc_traffic.json on the box so that a `bench` listed AFTER `pmc` reports this build's own counters
      python3 profiles/summarize_pmc.py $tag > /dev/null ;;
  esac
done
