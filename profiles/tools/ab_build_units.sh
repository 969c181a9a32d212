#!/bin/bash
# A/B builds with extra flags for SOME translation units: profiles/tools/ab_build_units.sh <name> "<unit> <unit> ..." <flags...>
#   -> cassierl_amd/lib/variants/libcassie2d_<name>.so  (the other units are the objects of the regular build; compare with tools/ab_bench.py, load with CASSIE2D_LIB)
# The units keep the regular build's per-unit flags (cassierl_amd/build.py UNIT_FLAGS: -ffp-contract=on for the leg / duo units -- the bit-identity
# between kernels depends on it) and go through the ISA guard; the extra flags come on top.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root" && python3 -m cassierl_amd.build --variant "$@"
