#!/bin/bash
# A/B builds of the Cassie3d translation unit with extra flags: profiles/tools/ab_build_3d.sh <name> <flags...> -> _ab/lib3d_<name>.so (load it with CASSIE2D_LIB=...)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/_ab
name=$1; shift
cd $root/cassierl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c -o $root/_ab/tu_3d_$name.o tu_3d.hip
objs=$(ls $root/cassierl_amd/lib/obj/*.o | grep -v tu_3d.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/_ab/lib3d_$name.so $objs $root/_ab/tu_3d_$name.o
echo built _ab/lib3d_$name.so
