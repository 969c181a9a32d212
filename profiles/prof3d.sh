#!/bin/bash
# Cassie3d kernel-trace + two PMC passes (run on the GPU box from the repo root: bash profiles/prof3d.sh [tag])
set -u
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/${1:-r05_3d}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats3d -o c3 -- python3 $root/tools/bench_cassie3d.py --steps 20 > $out/stats3d.log 2>&1
find $out/stats3d -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-200
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc3d_sq1 -o pmc -- python3 $root/tools/bench_cassie3d.py --steps 4 > $out/pmc3d_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/pmc3d_sq2 -o pmc -- python3 $root/tools/bench_cassie3d.py --steps 4 > $out/pmc3d_sq2.log 2>&1
python3 - "$out" <<'P'
import csv,glob,collections,sys
for d in ("pmc3d_sq1","pmc3d_sq2"):
    f=glob.glob("%s/%s/**/*counter_collection.csv"%(sys.argv[1],d), recursive=True)
    if not f: print(d,"no file"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVES","SQ_WAIT_ANY"): cnt[k]+=1
    for k,v in acc.items():
        if "step3d" in k: print(d,k,cnt[k],{a:b/max(1,cnt[k]) for a,b in v.items()})
P
