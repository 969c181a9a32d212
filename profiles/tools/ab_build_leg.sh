#!/bin/bash
# A/B builds of the leg translation unit with extra flags: profiles/tools/ab_build_leg.sh <name> <flags...> -> _ab/libleg_<name>.so (compare with tests/ab_bench.py)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/_ab
name=$1; shift
cd $root/cassierl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c -o $root/_ab/tu_leg_$name.o tu_leg.hip
objs=$(ls $root/cassierl_amd/lib/obj/*.o | grep -v "tu_leg.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/_ab/libleg_$name.so $objs $root/_ab/tu_leg_$name.o
echo built _ab/libleg_$name.so
