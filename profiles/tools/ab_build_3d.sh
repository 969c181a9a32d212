#!/bin/bash
# A/B build of the Cassie3d translation unit with extra flags: profiles/tools/ab_build_3d.sh <name> <flags...> -> cassierl_amd/lib/variants/libcassie2d_<name>.so
set -e
name=$1; shift
exec "$(dirname "$0")/ab_build_units.sh" "$name" "tu_3d" "$@"
