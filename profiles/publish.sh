#!/bin/bash
# Copy the summaries of a GPU session (gpurun_out/<tag>/, written by profiles/run_gpu.sh and profiles/collect_pmc.sh) into
# profiles/ under the round tag.   usage: bash profiles/publish.sh <tag>
set -u
tag=$1
src=gpurun_out/$tag
[ -f $src/bench.json ] && tail -1 $src/bench.json > profiles/${tag}_bench.json
[ -f $src/fullsize.jsonl ] && cp $src/fullsize.jsonl profiles/${tag}_fullsize.jsonl
[ -f $src/pytest_gpu.log ] && tail -3 $src/pytest_gpu.log > profiles/${tag}_pytest_gpu.txt
[ -f $src/pytest_gpu_leg_forced.log ] && tail -3 $src/pytest_gpu_leg_forced.log > profiles/${tag}_pytest_gpu_leg_forced.txt
[ -f $src/trpo_65536.jsonl ] && cp $src/trpo_65536.jsonl profiles/${tag}_trpo_65536.jsonl
[ -f $src/trpo_walk_65536.jsonl ] && cp $src/trpo_walk_65536.jsonl profiles/${tag}_trpo_walk_65536.jsonl
[ -f $src/cassie3d_bench.json ] && cp $src/cassie3d_bench.json profiles/${tag}_cassie3d_bench.json
ks=$(find $src/stats -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$ks" ] && grep -v "at::native\|__amd_rocclr" "$ks" > profiles/${tag}_kernel_stats.csv
for w in osc fallen trpo 3d; do
  ks=$(find $src/stats_$w -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$ks" ] && grep -v "at::native\|__amd_rocclr" "$ks" > profiles/${tag}_kernel_stats_$w.csv
done
[ -d $src/pmc_pd_sq1 ] && python3 profiles/summarize_pmc.py $tag
ls -la profiles/${tag}_*
