#!/bin/bash
# A/B build of the two-lanes translation unit with extra flags: profiles/tools/ab_build_leg.sh <name> <flags...> -> cassierl_amd/lib/variants/libcassie2d_<name>.so
set -e
name=$1; shift
exec "$(dirname "$0")/ab_build_units.sh" "$name" "tu_leg" "$@"
