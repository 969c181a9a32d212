set -u
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out/${1:-r05_d}; mkdir -p $out
true
cd /tmp && export TMPDIR=/tmp
G1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
G2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE"
for d in 0 1; do
  export CASSIE2D_DUO=$d
  rocprofv3 --pmc $G1 --output-format csv -d $out/pmc_duo${d}_g1 -o pmc -- python3 $root/tools/prof_step.py 65536 4 PD > $out/pmc_duo${d}_g1.log 2>&1
  rocprofv3 --pmc $G2 --output-format csv -d $out/pmc_duo${d}_g2 -o pmc -- python3 $root/tools/prof_step.py 65536 4 PD > $out/pmc_duo${d}_g2.log 2>&1
done
python3 - "$out" <<'P'
import csv,glob,collections,sys
for d in ("pmc_duo0_g1","pmc_duo0_g2","pmc_duo1_g1","pmc_duo1_g2"):
    f=glob.glob("%s/%s/**/*counter_collection.csv"%(sys.argv[1],d), recursive=True)
    if not f: print(d,"no file"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:50]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,v in acc.items():
        if "env_step" in k: print(d,k,{a:round(b/max(1,cnt[(k,a)])) for a,b in v.items()}, "dispatches", max(cnt[(k,a)] for a in v))
P
