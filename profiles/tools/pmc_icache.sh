# Instruction-cache counters of the 64-environments-per-wavefront kernel (105 KB of code against a 64 KB instruction cache shared by two CUs).
# usage (GPU box): bash profiles/tools/pmc_icache.sh <tag>
set -u
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out/${1:-r05_k}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $out/avail.txt 2>&1
grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" $out/avail.txt | sort -u > $out/avail_sqc.txt
G1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
G2="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_VALU"
for d in 0 1; do
  export CASSIE2D_DUO=$d
  rocprofv3 --pmc $G1 --output-format csv -d $out/pmc_ic${d}_g1 -o pmc -- python3 $root/tools/prof_step.py 65536 4 PD > $out/pmc_ic${d}_g1.log 2>&1
  rocprofv3 --pmc $G2 --output-format csv -d $out/pmc_ic${d}_g2 -o pmc -- python3 $root/tools/prof_step.py 65536 4 PD > $out/pmc_ic${d}_g2.log 2>&1
done
python3 - "$out" <<'P'
import csv,glob,collections,sys
for d in ("pmc_ic0_g1","pmc_ic0_g2","pmc_ic1_g1","pmc_ic1_g2"):
    f=glob.glob("%s/%s/**/*counter_collection.csv"%(sys.argv[1],d), recursive=True)
    if not f: print(d,"no file"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:50]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,v in acc.items():
        if "env_step_leg" in k or "duo" in k: print(d,k,{a:round(b/max(1,cnt[(k,a)])) for a,b in v.items()}, "dispatches", max(cnt[(k,a)] for a in v))
P
cat $out/avail_sqc.txt | tr '\n' ' '
