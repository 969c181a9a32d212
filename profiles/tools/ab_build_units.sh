#!/bin/bash
# A/B builds with extra flags for SOME translation units: profiles/tools/ab_build_units.sh <name> "<unit> <unit> ..." <flags...>
#   -> _ab/lib_<name>.so  (the other units are the objects of the regular build; compare with tests/ab_bench.py, load with CASSIE2D_LIB)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/_ab
name=$1; units=$2; shift 2
cd $root/cassierl_amd/csrc
objs=$(ls $root/cassierl_amd/lib/obj/*.o)
for u in $units; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c -o $root/_ab/${u}_$name.o $u.hip &
  objs=$(echo "$objs" | grep -v "/$u.o")
done
wait
for u in $units; do objs="$objs $root/_ab/${u}_$name.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/_ab/lib_$name.so $objs
echo built _ab/lib_$name.so
