#!/bin/bash
# Hardware counters behind the roofline / issue-rate statements (MI355X box).   usage: bash profiles/collect_pmc.sh <tag>
# One rocprofv3 run per counter group, --pmc only (no tracing domains); FETCH_SIZE and WRITE_SIZE in separate passes as
# MI355X_MICROARCH.md prescribes.  Workloads: the bench workload at 65 536 envs (PD), configs[2] (OSC in the loop) at 65 536,
# configs[4] (Cassie3d) at 16 384 (PMC_WORKLOADS / PMC_GROUPS select a subset).  Raw CSVs land in gpurun_out/<tag>/pmc_*; profiles/summarize_pmc.py condenses them.
set -u
tag=${1:-r02_x}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
G1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
G3="SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES"   # r06: thread-level VALU activity (how many lanes of an instruction are EXEC-active)
G2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters, program args...
  local name=$1 ctrs=$2; shift 2
  rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc_$name" -o pmc -- python3 "$@" > "$out/pmc_$name.log" 2>&1
  echo "$name rc=$? $(find "$out/pmc_$name" -name '*counter_collection.csv' | head -1 | xargs -r wc -l)"
}
for wl in ${PMC_WORKLOADS:-pd osc c3}; do
  case $wl in
    pd)  prog="$root/tools/prof_step.py 65536 4 PD" ;;
    osc) prog="$root/tools/prof_step.py 65536 4 OSC" ;;
    c3)  prog="$root/tools/bench_cassie3d.py --envs 16384 --steps 4" ;;
  esac
  for grp in ${PMC_GROUPS:-sq1 sq2 sq3 fetch write}; do
    case $grp in
      sq1) run ${wl}_sq1 "$G1" $prog ;;
      sq2) run ${wl}_sq2 "$G2" $prog ;;
      sq3) run ${wl}_sq3 "$G3" $prog ;;
      fetch) run ${wl}_fetch "FETCH_SIZE" $prog ;;
      write) run ${wl}_write "WRITE_SIZE" $prog ;;
    esac
  done
done
