#!/bin/bash
# Hardware counters of the TRPO loop's matrix-core kernel (MI355X box): rocprofv3 --pmc only, one run per group.
#   usage: bash profiles/collect_pmc_trpo.sh <tag>   -> gpurun_out/<tag>/pmc_trpo_*/ ; the summary lines are printed
set -u
tag=${1:-r04_x}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for grp in a b; do
  case $grp in
    a) ctrs="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" ;;
    b) ctrs="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU" ;;
  esac
  rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc_trpo_$grp" -o pmc -- python3 "$root/tools/prof_fvp.py" 8 > "$out/pmc_trpo_$grp.log" 2>&1
  echo "trpo_$grp rc=$? $(find "$out/pmc_trpo_$grp" -name '*counter_collection.csv' | head -1 | xargs -r wc -l)"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pmc_trpo_*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trpo_kernel" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((r["Dispatch_Id"], f))
    for d, _ in seen: cnt[f] += 1
for k, c in acc.items():
    nd = max(cnt.values()) if cnt else 1
    print(k[:60], "dispatches", nd)
    for name, v in sorted(c.items()): print("   %-32s %.4g per dispatch" % (name, v / nd))
PY
