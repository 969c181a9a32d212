#!/bin/bash
# HBM traffic of the first physics tier against the batch size: FETCH_SIZE / WRITE_SIZE (separate --pmc passes, no tracing)
# of `tools/prof_step.py <n> 4 PD` for each library given, at 65 536 .. 524 288 envs.   usage: bash tools/pmc_size_sweep.sh <tag> lib.so [lib.so ...]
# Raw CSVs under gpurun_out/<tag>/; the summary (per-dispatch means after the first dispatch; FETCH_SIZE doubled, KiB -> bytes: summarize_pmc.py's
# conventions) goes to gpurun_out/<tag>/pmc_size_sweep.jsonl.
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; shift
out=$root/gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "$lib" .so)
  export CASSIE2D_LIB=$root/$lib
  for n in ${SWEEP_SIZES:-65536 131072 262144 524288}; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d "$out/pmc_${name}_${n}_$c" -o pmc -- python3 "$root/tools/prof_step.py" $n 4 PD > "$out/pmc_${name}_${n}_$c.log" 2>&1
      echo "$name $n $c rc=$?"
    done
  done
done
python3 - "$out" "$@" <<'P'
import csv, glob, json, os, sys
out, libs = sys.argv[1], sys.argv[2:]
rows = []
for lib in libs:
    name = os.path.basename(lib)[:-3]
    for n in (int(x) for x in os.environ.get("SWEEP_SIZES", "65536 131072 262144 524288").split()):
        r = dict(lib=name, n_envs=n)
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = glob.glob("%s/pmc_%s_%d_%s/**/*counter_collection.csv" % (out, name, n, c), recursive=True)
            if not f:
                continue
            per = {}
            for row in csv.DictReader(open(f[0])):
                if "env_step_duo_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    per.setdefault(row["Dispatch_Id"], 0.0)
                    per[row["Dispatch_Id"]] += float(row["Counter_Value"])
            v = [per[k] for k in sorted(per, key=int)][1:]
            if v:
                r[c + "_KiB_per_launch"] = sum(v) / len(v)
        if "FETCH_SIZE_KiB_per_launch" in r and "WRITE_SIZE_KiB_per_launch" in r:
            r["hbm_read_MB"] = round(2 * 1024 * r["FETCH_SIZE_KiB_per_launch"] / 1e6, 1)
            r["hbm_write_MB"] = round(1024 * r["WRITE_SIZE_KiB_per_launch"] / 1e6, 1)
            r["traffic_MB_per_launch"] = round(r["hbm_read_MB"] + r["hbm_write_MB"], 1)
            r["algorithmic_MB"] = round(905 * n / 1e6, 1)
            r["traffic_over_algorithmic"] = round(r["traffic_MB_per_launch"] / r["algorithmic_MB"], 1)
        rows.append(r)
with open(os.path.join(out, "pmc_size_sweep.jsonl"), "w") as f:
    for r in rows:
        f.write(json.dumps(r) + "\n"); print(json.dumps(r))
P
